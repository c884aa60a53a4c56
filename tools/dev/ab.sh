#!/bin/bash
# usage: ab.sh out libs...
out=$1; shift
for b in 4096 32768; do python tests/tools/lib_compare.py default118 newton $b 40 default "$@"; done > $out 2>&1
