#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
cp blackbox_amd/libbbx_hip.so /tmp/lib_orig.so
for v in ko6; do
  cp tools/exp/lib_$v.so blackbox_amd/libbbx_hip.so
  echo "== $v"; bash tools/exp/bkg_prof.sh | grep boxstats
done
cp /tmp/lib_orig.so blackbox_amd/libbbx_hip.so
