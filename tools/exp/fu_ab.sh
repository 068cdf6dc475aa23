cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in product fuold; do
  if [ $v = product ]; then unset BBX_LIB_PATH; else export BBX_LIB_PATH=$GRAFT_REPO_ROOT/tools/exp/_var/fuold/libbbx_hip.so; fi
  rm -rf gpurun_out/fu; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/fu -o r -- python3 tools/dbg/fu_time.py > gpurun_out/fu.log 2>&1
  python3 - $v <<PY
import csv,glob,sys
f=(glob.glob("gpurun_out/fu/*kernel_stats.csv")+glob.glob("gpurun_out/fu/*/*kernel_stats.csv"))[0]
for r in csv.DictReader(open(f)):
    if "funpack" in r["Name"]: print(sys.argv[1], r["Name"][:40], r["Calls"], round(float(r["AverageNs"])/1e3,1), "us  min", round(float(r["MinNs"])/1e3,1))
PY
done
rm -rf gpurun_out/fu
