#!/bin/bash
# Developer tool (round 2): cycle stamps of several diagnostic (W4_STAMP) builds.
for l in "$@"; do echo "=== $l"; GNLP_LIB_PATH=$PWD/$l timeout -k 10 200 python tools/w4_stamp.py 2>&1 | grep -v "per wave\|wave [0-9]:"; done
