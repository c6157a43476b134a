#!/bin/bash
# usage: tools/bench_brief.sh [bench args]: one line: fps, ms/step, per-class kernel ms
python bench.py --no-cpu-baseline "$@" 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print(d['value'], d['ms_per_step'], d['whole_forward']['kernel_ms_per_step'])"
