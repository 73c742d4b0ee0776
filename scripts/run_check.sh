set -u
mkdir -p gpurun_out/s3a
python -m pytest tests -m gpu -x -q > gpurun_out/s3a/pytest.log 2>&1; echo "pytest exit $?" >> gpurun_out/s3a/pytest.log
tail -5 gpurun_out/s3a/pytest.log
python bench.py > gpurun_out/s3a/bench.json 2> gpurun_out/s3a/bench.err; tail -c 2500 gpurun_out/s3a/bench.json
python scripts/measure_configs.py --configs 2,3,4,5 > gpurun_out/s3a/configs.jsonl 2>&1; cat gpurun_out/s3a/configs.jsonl
