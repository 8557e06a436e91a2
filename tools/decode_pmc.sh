#!/bin/bash
# (GPU box) fabric-side bytes per launch of the decode-layer kernels: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in SEPARATE passes over
# tools/decode_kernel_bench.py (counters in KiB; FETCH_SIZE doubled per MI355X_MICROARCH.md: gfx950 tallies 128-byte requests at 64 B).
OUT=${1:-gpurun_out/decode_pmc}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$ROOT/$OUT"
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$ROOT/$OUT/raw/$c" -- python3 "$ROOT/tools/decode_kernel_bench.py" > "$ROOT/$OUT/bench_$c.log" 2>&1
done
cd "$ROOT"
python3 tools/pmc_summary.py "$OUT/raw" decode_ > "$OUT/pmc_decode.json"
python3 tools/pmc_summary.py "$OUT/raw" attn_decode > "$OUT/pmc_attn.json"
python3 - "$OUT" <<'PY'
import json, sys
out = sys.argv[1]
d = {}
for f in ("pmc_decode.json", "pmc_attn.json"):
    d.update(json.load(open(f"{out}/{f}")))
alg = {"decode_ln_qkv_fc1_lds_kernel<2, 4>": 2.0 * 7 * 1024 * 1024 + 32 * 1024 * 4, "decode_out_lds_kernel<2>": 2.0 * 5 * 1024 * 1024 + 32 * 5120 * 2,
       "attn_decode_flat_kernel<64, 10>": 32 * 294 * 2 * 1024 * 2.0, "decode_head_kernel<2>": 2.0 * 50304 * 1024}
res = {"method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python3 tools/decode_kernel_bench.py; per launch; FETCH_SIZE in KiB doubled (gfx950), fabric side of the L2 (Infinity-Cache hits are counted)", "kernels": {}}
for k, c in d.items():
    fe, wr = c.get("FETCH_SIZE"), c.get("WRITE_SIZE")
    a = next((v for n, v in alg.items() if k.startswith(n.split("<")[0]) and n == k), None)
    res["kernels"][k] = {"fetch_MB_x2": None if fe is None else round(fe * 2 * 1024 / 1e6, 2), "write_MB": None if wr is None else round(wr * 1024 / 1e6, 2),
                         "algorithmic_read_MB": None if a is None else round(a / 1e6, 2)}
json.dump(res, open(f"{out}/decode_pmc.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
rm -rf "$OUT/raw"
