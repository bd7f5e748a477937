#!/bin/bash
# last check of the round: new tests, smoke(), the default bench line
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -x -q -k "clip_b32 or resnet or from_images or launch_variants" 2>&1 | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 300 python bench.py --steps 1 --warmup 1 --cpu-seconds 3 2>/dev/null | python -c "
import json,sys; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', b['value'], b['ms_per_step'], b['roofline']['frac'], b['cpu_baseline']['value'])"
