#!/bin/bash
# words of real text (4096 x 4096 tiny strings) and the headline, default library vs build variants
cd "$(dirname "$0")/.."; ROOT=$PWD; OUT=$ROOT/gpurun_out/real_text; mkdir -p "$OUT"
cat SURVEY.md DESIGN.md PAPERS.md SNIPPETS.md INTEGRATION.md /opt/skills/guides/*.md > "$OUT/corpus.txt" 2>/dev/null
for v in default "$@"; do
  if [ $v = default ]; then unset STRINGZILLAS_ROCM_LIBRARY; else export STRINGZILLAS_ROCM_LIBRARY=$ROOT/stringzilla_amd/lib_variants/$v/libstringzillas_rocm_shared.so; fi
  echo "== $v"
  for engine in levenshtein levenshtein_utf8; do python scripts/measure_dataset.py --dataset "$OUT/corpus.txt" --tokens words --engine $engine --queries 4096 --candidates 4096 | cut -c60-230; done
  python bench.py --config 2 --extra-configs none --no-cpu-baseline --steps 300 --warmup 30 | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg2', l['value'], l['ms_per_step'], l['roofline']['kernel_ms'], l['results_checksum'])"
done
