#!/bin/bash
# build an A/B variant of the library from the same sources: tools/ab_build.sh <name> [-DFLAG ...]  ->  tools/ab/<name>.so
# (compare on one box with tools/ab_bench.sh <name1>.so <name2>.so); the flags go to every translation unit
cd "$(dirname "$0")/.." && mkdir -p tools/ab
n=$1; shift
python3 - "$n" "$@" <<'PY'
import sys
from transport_se_amd import _lib
print(_lib.build(out="tools/ab/%s.so" % sys.argv[1], flags=sys.argv[2:]))
PY
