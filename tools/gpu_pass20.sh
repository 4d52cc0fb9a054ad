#!/bin/bash
for i in 1 2 3 4; do
timeout 600 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-pmc 2>&1 | grep -v amdgpu.ids | grep "ranked runner\|Assertion\|value" | cut -c1-400
done
