#!/bin/bash
# same lib, env var toggle
for b in 4096 32768; do
  PPN_NO_REBALANCE=1 python tests/tools/lib_compare.py default118 newton $b 40 default
  python tests/tools/lib_compare.py default118 newton $b 40 default
done
