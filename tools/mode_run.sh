#!/bin/bash
# how often does a fresh process land in the slow mode of the traversal kernel?  usage: mode_run.sh <repeats>
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
B=nvdiffrecmc_amd/csrc/build
cp $B/libnvdr_hip.so /tmp/libnvdr_hip.so.orig
for f in /tmp/libnvdr_hip.so.orig $B/variants/libnvdr_hip.so.*; do
  cp $f $B/libnvdr_hip.so; echo "== $f"
  for i in $(seq 1 ${1:-8}); do PROBE_VIEWS=1 timeout 120 python tools/stage_probe.py 8,6,6 2>&1 | grep "fwd gen" | sed 's/.*trace \([0-9.]*\) shade.*/\1/' | tr '\n' ' '; done; echo
done
cp /tmp/libnvdr_hip.so.orig $B/libnvdr_hip.so
