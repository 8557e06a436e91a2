"""Per-shape L2-fabric traffic of the GEMM kernels (tools/pmc_traffic.sh) against (a) the algorithmic bytes and (b) what the XCD-aware tile
order predicts for eight non-coherent L2s: every XCD fetches the operand panels of ITS tiles itself, so the fabric-side read count is the sum
over XCDs and rounds of the panel footprints, served by the Infinity Cache after the first XCD (FETCH_SIZE counts those hits too,
MI355X_MICROARCH.md 'HBM')."""
import csv
import json
import sys

root = sys.argv[1]
M = 9216
SH = {  # name: (M, N, K, tile_m, tile_n, C bytes/elt, extra epilogue read B/elt, extra epilogue write B/elt)
    "qkv": (M, 3072, 1024, 144, 256, 2, 0, 0), "dense": (M, 1024, 1024, 144, 256, 2, 0, 0), "fc1": (M, 4096, 1024, 144, 256, 2, 0, 2),
    "fc2": (M, 1024, 4096, 144, 256, 4, 6, 0), "dfc2": (M, 4096, 1024, 144, 256, 2, 2, 0), "dfc1": (M, 1024, 4096, 144, 256, 2, 0, 0),
    "dqkv": (M, 1024, 3072, 144, 256, 2, 0, 0), "dao": (M, 1024, 1024, 144, 256, 2, 0, 0),
}


def per_launch(path, counter):
    tot, n = 0.0, 0
    with open(path) as fp:
        for row in csv.DictReader(fp):
            if row["Counter_Name"] == counter and ("gemm_pp_kernel" in row["Kernel_Name"] or "gemm_z_kernel" in row["Kernel_Name"]
                                                   or "gemm_bf16" in row["Kernel_Name"]):
                tot += float(row["Counter_Value"]); n += 1
    return (tot / n if n else None), n


def predicted_fabric_reads(Mr, N, K, tm, tn):
    """Tile id = round * 256 + slot; slot -> XCD = slot % 8, the 32 slots of an XCD form a GROUP_M x (32 / GROUP_M) patch of tiles
    (gemm_pp_launch): an XCD's L2 (4 MiB) sees GROUP_M A panels and 32 / GROUP_M B panels per round and keeps nothing across rounds."""
    tiles_n = N // tn
    gm = 4 if tiles_n >= 8 else (8 if tiles_n >= 4 else 16)
    gn = min(32 // gm, tiles_n)
    tiles = (Mr // tm) * tiles_n
    rounds = -(-tiles // 256)
    per_xcd_round = (gm * tm + gn * tn) * K * 2
    return 8 * rounds * per_xcd_round


out = {"method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python3 tools/gemm_shape_run.py <shape> 10: isolated launches "
                 "of each product with the step's epilogue, automatic dispatch; counters in KiB; FETCH_SIZE doubled (gfx950 tallies 128-B requests at "
                 "64 B, MI355X_MICROARCH.md); FETCH_SIZE is the L2's fabric side: Infinity-Cache hits are counted",
       "shapes": {}}
tot_alg = tot_meas = 0.0
for name, (Mr, N, K, tm, tn, cb, er, ew) in SH.items():
    f, n = per_launch(f"{root}/{name}_FETCH_SIZE.csv", "FETCH_SIZE")
    w, _ = per_launch(f"{root}/{name}_WRITE_SIZE.csv", "WRITE_SIZE")
    if f is None:
        continue
    alg_r = (Mr * K + N * K) * 2 + Mr * N * er
    alg_w = Mr * N * (cb + ew)
    pred = predicted_fabric_reads(Mr, N, K, tm, tn) + Mr * N * er
    e = {"launches": n, "fetch_MB_x2": round(f * 1024 * 2 / 1e6, 1), "write_MB": round(w * 1024 / 1e6, 1),
         "algorithmic_read_MB": round(alg_r / 1e6, 1), "algorithmic_write_MB": round(alg_w / 1e6, 1),
         "predicted_fabric_read_MB_8_private_L2": round(pred / 1e6, 1)}
    e["fetch_over_algorithmic"] = round(e["fetch_MB_x2"] / e["algorithmic_read_MB"], 2)
    e["fetch_over_predicted"] = round(e["fetch_MB_x2"] / e["predicted_fabric_read_MB_8_private_L2"], 2)
    e["total_over_algorithmic"] = round((e["fetch_MB_x2"] + e["write_MB"]) / (e["algorithmic_read_MB"] + e["algorithmic_write_MB"]), 2)
    out["shapes"][name] = e
    tot_alg += alg_r + alg_w
    tot_meas += (e["fetch_MB_x2"] + e["write_MB"]) * 1e6
f, n = per_launch(f"{root}/wgrp2_FETCH_SIZE.csv", "FETCH_SIZE")
w, _ = per_launch(f"{root}/wgrp2_WRITE_SIZE.csv", "WRITE_SIZE")
if f is not None:
    alg_r = 2 * (M * (1024 + 4096 + 4096 + 1024 + 3072 + 1024 + 1024 + 1024) * 2 + 12 * 1024 * 1024 * 4)   # operands + the fp32 gradients it accumulates into
    alg_w = 2 * 12 * 1024 * 1024 * 4
    out["shapes"]["wgrp2 (grouped dW of two layers, 128 x 256 tiles)"] = {
        "launches": n, "fetch_MB_x2": round(f * 1024 * 2 / 1e6, 1), "write_MB": round(w * 1024 / 1e6, 1), "algorithmic_read_MB": round(alg_r / 1e6, 1),
        "algorithmic_write_MB": round(alg_w / 1e6, 1), "fetch_over_algorithmic": round(f * 1024 * 2 / alg_r, 2)}
# launch mix of one 410M MAFED step (student forward + backward, 22-layer teacher forward; bench.py `kernels`): the per-launch average the
# bench line's roofline.traffic quotes
MIX = {"qkv": 46, "dense": 46, "dao": 24, "fc1": 46, "fc2": 46, "dfc2": 24, "dfc1": 24, "dqkv": 24}
num = den = alg = 0.0
for k, n in MIX.items():
    e = out["shapes"].get(k)
    if e:
        num += n * (e["fetch_MB_x2"] + e["write_MB"]); alg += n * (e["algorithmic_read_MB"] + e["algorithmic_write_MB"]); den += n
for k, e in out["shapes"].items():
    if k.startswith("wgrp2"):
        num += 12 * (e["fetch_MB_x2"] + e["write_MB"]); alg += 12 * (e["algorithmic_read_MB"] + e["algorithmic_write_MB"]); den += 12
if den:
    out["hbm_MB_per_launch"] = round(num / den, 1)
    out["algorithmic_MB_per_launch"] = round(alg / den, 1)
    out["launch_mix_per_step"] = dict(MIX, wgrp2=12)
    out["reading"] = ("fetch_over_predicted ~ 1.0-1.1: the fabric-side reads are what eight private L2s must fetch for their own tiles (every XCD reads the "
                      "A panels of its GROUP_M tile rows and the B panels of its 32 / GROUP_M tile columns once per round); the first XCD's fetch comes "
                      "from HBM, the other seven hit the Infinity Cache (operands of one product: 21-141 MB < 256 MiB), and FETCH_SIZE counts both. "
                      "Writes equal the algorithmic bytes (no partial-line stores).")
print(json.dumps(out, indent=1))
