#!/bin/bash
# Same-box A/B of library builds on tools/ktimes.py cases:  bash tools/ab_libs.sh "CASE CASE ..." lib1.so lib2.so ...  [ROUNDS=2]
# Alternates the libraries ROUNDS times so that box drift shows up as spread inside a library, not as a difference.
CASES=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for round in $(seq 1 ${ROUNDS:-2}); do
  for lib in "$@"; do
    BFSM_LIB=$R/$lib timeout -k 10 300 python3 $R/tools/ktimes.py $CASES || exit 1
  done
done
