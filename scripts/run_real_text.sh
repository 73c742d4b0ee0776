#!/bin/bash
# run_real_text.sh - the reference's benchmark on REAL text (bench/shared.hpp:240-290): the prose that ships with this
# repository and with the image (no dataset can be downloaded here), tokenised as words and as lines, 4096 x 4096 tokens.
ROOT=$(cd "$(dirname "$0")/.." && pwd); OUT=$ROOT/gpurun_out/real_text; mkdir -p "$OUT"; cd "$ROOT"
# (the design notes of rounds 1-4 stand in for DESIGN.md, which round 5 rewrote: the corpus stays the one of the earlier rounds)
cat SURVEY.md docs/history/DESIGN_rounds_1_to_4.md PAPERS.md SNIPPETS.md INTEGRATION.md /opt/skills/guides/*.md > "$OUT/corpus.txt" 2>/dev/null
ls -la "$OUT/corpus.txt"
for tokens in words lines; do for engine in levenshtein levenshtein_utf8; do
  python scripts/measure_dataset.py --dataset "$OUT/corpus.txt" --tokens $tokens --engine $engine --queries 4096 --candidates 4096 | tee -a "$OUT/real_text.jsonl"
done; done
