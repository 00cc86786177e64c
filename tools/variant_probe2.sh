#!/usr/bin/env bash
# count-kernel time of tuning builds at 1e8 reads: args = name:target ...
cd "$(dirname "$0")/.."
for v in "$@"; do
  lib=${v%%:*}; tgt=${v##*:}
  echo -n "== $lib target=$tgt: "
  SNK_LIB_PATH=$PWD/supernova_amd/variants/libsnk_$lib.so SNK_TARGET_INST=$tgt timeout 300 python tools/msp_probe.py 1e8 0 2>&1 | tail -1
done
