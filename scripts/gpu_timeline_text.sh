#!/bin/bash
# gpu_timeline_text.sh TOKENS ENGINE - kernel timeline of one call on the real-text batch (scripts/run_real_text.sh makes the corpus)
cd "$(dirname "$0")/.."; export TMPDIR=/tmp
OUT=gpurun_out/timeline_text; mkdir -p $OUT gpurun_out/real_text; rm -rf $OUT/*
cat SURVEY.md DESIGN.md PAPERS.md SNIPPETS.md INTEGRATION.md /opt/skills/guides/*.md > gpurun_out/real_text/corpus.txt 2>/dev/null
rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python scripts/measure_dataset.py --dataset gpurun_out/real_text/corpus.txt --tokens ${1:-words} --engine ${2:-levenshtein_utf8} --queries 4096 --candidates 4096 > $OUT/run.log 2>&1
python scripts/kernel_timeline.py $(find $OUT -name "*kernel_trace.csv" | head -1) ${3:-60}
