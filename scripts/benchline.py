import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print("value", d["value"], "ms/step", d["ms_per_step"], "kernel_ms", d["roofline"]["kernel_ms"], "kernel_gcups", d["roofline"]["kernel_gcups"], "checksum", d["results_checksum"])
