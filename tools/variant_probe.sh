#!/usr/bin/env bash
# run the 1e7 probe for every tuning build: prints the count-phase time
cd "$(dirname "$0")/.."
for v in "$@"; do
  lib=${v%%:*}; tgt=${v##*:}
  echo "== $lib target=$tgt"
  SNK_LIB_PATH=$PWD/supernova_amd/variants/libsnk_$lib.so SNK_TARGET_INST=$tgt python tools/scale_probe.py 1e7 2>&1 | grep -E "phases|n=" | tail -2
done
