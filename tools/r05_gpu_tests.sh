#!/bin/bash
# development tool (round 5), on the GPU box: a list of pytest selections, each in ONE process under its own `timeout -k`, verbose
# (a line per test as it starts: a hung test is named in the log), the next selection only if the one before was not killed.
#   bash tools/r05_gpu_tests.sh LOGNAME "selection 1" "selection 2" ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
LOG=$R/gpurun_out/$1
shift
cd $R
: > $LOG
for SEL in "$@"; do
  echo "=== pytest $SEL" >> $LOG
  timeout -k 10 ${TEST_TIMEOUT:-900} python3 -u -m pytest $SEL -v -m gpu -o faulthandler_timeout=${FH_TIMEOUT:-150} -p no:cacheprovider >> $LOG 2>&1
  rc=$?
  echo "=== rc=$rc" >> $LOG
  if [ $rc -ge 124 ]; then echo "killed: stopping here" >> $LOG; tail -n 40 $LOG; exit $rc; fi
done
grep -E "^(=== |FAILED|ERROR)|passed|failed" $LOG | tail -n 40
