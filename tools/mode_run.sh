#!/bin/bash
# how often does a fresh process land in the slow mode of the traversal kernel?  usage: mode_run.sh <repeats>
# prints per process: one-view traversal ms | shader clock (MHz) and XCD mask seen by the counting launch
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
B=nvdiffrecmc_amd/csrc/build
cp $B/libnvdr_hip.so /tmp/libnvdr_hip.so.orig
for f in /tmp/libnvdr_hip.so.orig $(ls $B/variants/libnvdr_hip.so.* 2>/dev/null); do
  cp $f $B/libnvdr_hip.so; echo "== $f"
  for i in $(seq 1 ${1:-8}); do PROBE_VIEWS=1 timeout 120 python tools/stage_probe.py 8,6,6 2>&1 | grep "fwd gen\|shader clock" | sed 's/.*fwd gen [0-9.]* trace \([0-9.]*\) shade.*/\1/; s/shader clock during the counting launch: //' | tr '\n' ' '; echo; done
done
cp /tmp/libnvdr_hip.so.orig $B/libnvdr_hip.so
