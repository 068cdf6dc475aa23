#!/bin/bash
# GPU box: (1) the C test against torch's bundled HIP runtime (through soname symlinks); (2) torch with the SYSTEM runtime preloaded
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
TL=$(python3 -c "import torch,os;print(os.path.join(os.path.dirname(torch.__file__),'lib'))")
mkdir -p gpurun_out/d2h /tmp/trt; : > gpurun_out/d2h/rt.txt
ln -sf $TL/libamdhip64.so /tmp/trt/libamdhip64.so.7; ln -sf $TL/libhsa-runtime64.so /tmp/trt/libhsa-runtime64.so.1
ln -sf $TL/libamd_comgr.so /tmp/trt/libamd_comgr.so.3; ln -sf $TL/librocprofiler-register.so /tmp/trt/librocprofiler-register.so.0
run() { tag=$1; shift
  rm -rf gpurun_out/d2h/f_$tag
  env "$@" > gpurun_out/d2h/fo_$tag.txt 2>&1
  echo "== $tag: $(grep -E 'D2H|Error|error' gpurun_out/d2h/fo_$tag.txt | head -3)" >> gpurun_out/d2h/rt.txt
  f=$(ls gpurun_out/d2h/f_$tag/*/*kernel_stats.csv gpurun_out/d2h/f_$tag/*kernel_stats.csv 2>/dev/null | head -1)
  [ -n "$f" ] && grep -i "copyBuffer" $f | cut -c1-90 >> gpurun_out/d2h/rt.txt
  rm -rf gpurun_out/d2h/f_$tag
}
LD_LIBRARY_PATH=/tmp/trt ldd ./tools/exp/d2h_flags | grep -E "hip|hsa" >> gpurun_out/d2h/rt.txt
run c_torchrt LD_LIBRARY_PATH=/tmp/trt timeout -k 10 60 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/d2h/f_c_torchrt -o r -- ./tools/exp/d2h_flags 0
run py_sysrt LD_PRELOAD=/opt/rocm/lib/libamdhip64.so.7:/opt/rocm/lib/libhsa-runtime64.so.1 timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/d2h/f_py_sysrt -o r -- python3 tools/exp/d2h_path.py torch
cat gpurun_out/d2h/rt.txt; tail -5 gpurun_out/d2h/fo_py_sysrt.txt
