#!/bin/bash
# (GPU box) rocprofv3 kernel statistics of the validation-decode bench -> $OUT/decode_kernel_stats.csv (rows of the decode kernels)
OUT=${1:-gpurun_out/decode_prof}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$ROOT/$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/raw" -- python3 "$ROOT/tools/decode_bench.py" > "$ROOT/$OUT/decode_bench.log" 2>&1
cd "$ROOT"
ks=$(find "$OUT/raw" -name "*kernel_stats.csv" | head -1)
{ echo "# rocprofv3 --kernel-trace --stats --output-format csv -- python3 tools/decode_bench.py   (1x MI355X; all four decode forms run in this bench)"; head -1 "$ks"; grep -E "decode_|attn_decode|rotate_k|gemm_skinny|layernorm_fwd" "$ks" | sed 's/(mafed::[A-Za-z]*Args)//; s/void mafed:://'; } | cut -c1-260 > "$OUT/decode_kernel_stats.csv"
rm -rf "$OUT/raw"
cat "$OUT/decode_kernel_stats.csv"
