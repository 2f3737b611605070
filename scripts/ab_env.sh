#!/bin/bash
# Same-box A/B of an environment switch on a command: VAR=name, A/B = the two values, CMD = command
set -o pipefail
for r in 1 2; do
  for v in "$A" "$B"; do
    echo -n "$VAR=$v: "
    env $VAR=$v timeout -k 10 300 $CMD || exit 1
  done
done
