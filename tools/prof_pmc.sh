#!/bin/bash
# GPU box: the two PMC passes (own runs, no trace domains) -> gpurun_out/pmc_traffic.json
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
rm -rf gpurun_out/pmc_f gpurun_out/pmc_w
timeout -k 10 500 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_f -o r -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-extras --lanes 1 --depth 2 > gpurun_out/pmc_f.log 2>&1 || exit 1
timeout -k 10 500 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_w -o r -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-extras --lanes 1 --depth 2 > gpurun_out/pmc_w.log 2>&1 || exit 1
python3 tools/pmc_traffic.py gpurun_out/pmc_f gpurun_out/pmc_w > gpurun_out/pmc_traffic.json && rm -rf gpurun_out/pmc_f gpurun_out/pmc_w
python3 - <<PY
import json
d=json.load(open('gpurun_out/pmc_traffic.json'))['kernels']
for k in ('k_calibrate','k_lac_cand','k_final_rows','k_img_cols','k_img_rows','k_psf_cols','k_psf_rows','k_var_cols','k_cols_fwd','k_bkg_boxstats','k_spline_zoom'):
    if k in d: print(k, round(d[k]['traffic_bytes_per_launch']/1e6,1),'MB')
PY
