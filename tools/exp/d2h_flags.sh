#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
[ -x tools/exp/d2h_flags ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o tools/exp/d2h_flags tools/exp/d2h_flags.hip
TL=$(python3 -c "import torch,os;print(os.path.join(os.path.dirname(torch.__file__),'lib'))")
mkdir -p gpurun_out/d2h; : > gpurun_out/d2h/flags.txt
run() { tag=$1; shift
  rm -rf gpurun_out/d2h/f_$tag
  env "$@" > gpurun_out/d2h/fo_$tag.txt 2>&1
  echo "== $tag: $(grep 'D2H' gpurun_out/d2h/fo_$tag.txt)" >> gpurun_out/d2h/flags.txt
  f=$(ls gpurun_out/d2h/f_$tag/*/*kernel_stats.csv gpurun_out/d2h/f_$tag/*kernel_stats.csv 2>/dev/null | head -1)
  if [ -n "$f" ]; then grep -i "copyBuffer" $f | cut -c1-100 >> gpurun_out/d2h/flags.txt; else echo "  (no kernel stats: no kernels ran)" >> gpurun_out/d2h/flags.txt; fi
  rm -rf gpurun_out/d2h/f_$tag
}
run sysrt A=1 timeout -k 10 60 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/d2h/f_sysrt -o r -- ./tools/exp/d2h_flags 0
run sysrt_prio A=1 timeout -k 10 60 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/d2h/f_sysrt_prio -o r -- ./tools/exp/d2h_flags 0 prio
run torchrt LD_LIBRARY_PATH=$TL timeout -k 10 60 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/d2h/f_torchrt -o r -- ./tools/exp/d2h_flags 0
run torchrt_prio LD_LIBRARY_PATH=$TL timeout -k 10 60 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/d2h/f_torchrt_prio -o r -- ./tools/exp/d2h_flags 0 prio
ldd ./tools/exp/d2h_flags | grep -i hip >> gpurun_out/d2h/flags.txt
cat gpurun_out/d2h/flags.txt
