#!/bin/bash
# Same-box A/B of BINARIES (round-4 review item 6): the round-3 tree (commit 62bff04, exported to ab_r03/ and built there -- see
# profiles/r05/README) against HEAD, alternating, 7B, 1024 timed greedy steps each, everything else off.  usage: tools/ab_r03.sh [pairs] > out.txt
cd "$(dirname "$0")/.."
R=$PWD; N=${1:-3}
FLAGS="--steps 1024 --warmup 32 --no-cpu-baseline --ref-steps 0 --prefill-chunks 0 --config2-steps 0"
one() {   # $1 = label, $2 = tree
  ( cd $2 && timeout 300 python bench.py $FLAGS 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
k=d['kernels']
print('$1  %.2f tok/s  %.4f ms/step  ' % (d['value'], d['ms_per_step']) + '  '.join('%s %.2f' % (n, k[n]['us']) for n in ('att_kvr_wkv','att_out','ffn_rk','ffn_v','head') if n in k))" )
}
echo "# 7B, 1024 timed greedy steps after 32 warm-up steps, alternating on ONE box: r03 = commit 62bff04's csrc + bench.py, HEAD = this tree"
for i in $(seq $N); do one "r03 " $R/ab_r03; one "HEAD" $R; done
