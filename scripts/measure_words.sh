#!/bin/bash
# measure_words.sh [TAG] - the tiny-token regime from plain C (tests/native/words_probe.c): parity on edge-shaped batches, then
# 4096 x 4096 words of this repository's prose timed as a stream of fresh batches.  Output: gpurun_out/TAG/words.txt
ROOT=$(cd "$(dirname "$0")/.." && pwd); TAG=${1:-words}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"; cd "$ROOT"
P=tests/native/bin/words_probe
CORPUS=scratch/corpus.txt
[ -f $CORPUS ] || { mkdir -p scratch; cat SURVEY.md docs/history/DESIGN_rounds_1_to_4.md PAPERS.md SNIPPETS.md INTEGRATION.md /opt/skills/guides/*.md > $CORPUS 2>/dev/null; }
{
# (`tiny` = 2: the launch scores dense mixes too instead of refusing them - the testing knob; = 1: it refuses a block or span of which more than a quarter is long)
for shape in "mix:60:40 100 700 3" "mix:100:255 300 1000 3 7" "mix:180:70 200 520 3" "mix:300:128 65 260 3" "mix:0:16 33 257 3" "mix:1000:200 40 300 2" "mix:20:64 1 1 2" "mix:500:33 257 31 3 1"; do
  echo "== SZS_ROCM_TINY=2 $shape"; SZS_ROCM_TINY=2 timeout 120 $P $shape 2>&1 | tail -4
done
for shape in "mix:100:255 300 1000 3 7" "mix:1000:200 40 300 2"; do echo "== SZS_ROCM_TINY=1 $shape"; SZS_ROCM_TINY=1 timeout 120 $P $shape 2>&1 | tail -1; done
echo "== wide"; PROBE_WIDE=1 SZS_ROCM_TINY=1 timeout 120 $P mix:50:64 37 513 3 87 2>&1 | tail -3
echo "== words, automatic"; timeout 200 $P file:$CORPUS 4096 4096 8 2>&1 | tail -5
echo "== words, trace"; SZS_ROCM_TRACE=1 PROBE_NO_ORACLE=1 timeout 200 $P file:$CORPUS 4096 4096 4 2>&1 | grep -v "^run" | tail -6
echo "== words, the ordinary kernels (SZS_ROCM_TINY=0)"; SZS_ROCM_TINY=0 PROBE_NO_ORACLE=1 timeout 200 $P file:$CORPUS 4096 4096 8 2>&1 | tail -1
echo "== synthetic words mix:0:16"; PROBE_NO_ORACLE=1 timeout 200 $P mix:0:16 4096 4096 8 2>&1 | tail -1
} > "$OUT/words.txt" 2>&1
cat "$OUT/words.txt"
