#!/bin/bash
# one-off: hunt the rare failing search (cold posting cache, 64 threads, detailed scores, many distinct queries)
for i in 1 2 3 4 5 6 7 8; do
RB_DISTINCT_QUERIES=3072 RB_DETAILED=1 timeout 300 tools/bin/ranked_bench 10000000 200000 3 16 64 > /tmp/rb_$i.out 2> /tmp/rb_$i.err; echo "run $i rc=$? $(grep -c qps /tmp/rb_$i.out)"; grep -v "^$" /tmp/rb_$i.err | grep -v "posting_cache\|^{" | tail -3 | cut -c1-400
done
