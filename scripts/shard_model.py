"""Cost model of the row-sharded filter in the exact configuration, from the one-GPU measurements under
profiles/: python scripts/shard_model.py [round-prefix, default r04]

Inputs  : profiles/<r>_bench_n2000_f32x.json, <r>_bench_n5000_f32x.json (ms per frame, downdate launches, sweep time, panels, and
          the time between the end of a sweep and the start of its downdate = dx, state update, triangular inverse, int8 GEMM);
          profiles/<r>_kernel_stats_n5000_f32x.csv only for the RATIO GEMM : inverse inside that interval (the trace includes the
          first frame twice, so its totals are not per-frame figures of the timed frames).
Model   : downdate x 2/G (a rank computes both triangles of its own rows); rows of B / G with the sweep bounded below by the
          dependent chain (CHAIN_US per 32-row panel, the fast configuration's measured launch period); the GEMM / G; the inverse
          and everything else replicated; per update a rank receives (G-1)/G of the five digit planes of B (5 B per element) and of the
          columns of S (8 m^2 B; until round 4, above 2048 rows: of the rows of G, 8 B per element) over G-1 xGMI links of LINK_GBS each, not overlapped with compute.
Round 6: matching and the RANSAC hypotheses are divided by feature ownership (csrc/engine.cpp: match_sharded_dev, ransac_dev): their
stage times / G, and every exchange of a frame is charged a LATENCY (XCHG_LAT_US each: the grouped send / recv is enqueued on the
engine's stream; 2 x S_i after the predictions, 3 match tables, 2 per RANSAC batch, and per update the diagonal of P, the columns of
S and the digit planes) -- rounds 3-5 priced the bytes only.
No multi-GPU measurement exists on this pool: this is arithmetic on one-GPU measurements, not a scaling result."""
import csv
import json
import os
import sys

CHAIN_US = 9.1    # us per panel of the dependent chain (k_chol_step with the rows-of-B role hidden; profiles/r04_bench_n1000_f32.json)
LINK_GBS = 153.0  # one xGMI link, one direction (MI355X_MICROARCH.md)
XCHG_LAT_US = 15.0  # per exchange: launch + rendezvous of a grouped send / recv on the stream (assumed; not measurable on one GPU)
SYM_ABOVE = True  # G by symmetry also above 2048 rows (csrc/kernels_update.hip, sym_g): no rows of G travel at any size
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles")


def kernel_ms(path, needles, frames):
    tot = 0.0
    if not os.path.exists(path):
        return None
    with open(path) as f:
        for row in csv.DictReader(f):
            name = row.get("Name") or row.get("kernel")
            if any(n in name for n in needles):
                tot += float(row["TotalDurationNs"]) / 1e6 if "TotalDurationNs" in row else float(row["total_ms"])
    return tot / frames


def per_frame_ms(path, needles):
    """ms per frame of the kernels whose name contains one of `needles`, from a kernel-trace summary: the trace's frames = its k_match
    (or k_ncc_match) calls"""
    if not os.path.exists(path):
        return 0.0
    rows = list(csv.DictReader(open(path)))
    frames = sum(int(r["Calls"]) for r in rows if "k_match(" in r["Name"] or "k_ncc_match" in r["Name"]) or 1
    return sum(float(r["TotalDurationNs"]) for r in rows if any(n in r["Name"] for n in needles)) / 1e6 / frames


def model(tag, d, n_state, gemm_share, ranks, owner_kernels_ms=0.0):
    t1 = d["ms_per_step"]
    roof, sw = d["roofline"], d["roofline_sweep"]
    steps = d["steps"]
    down = roof["avg_launch_ms"] * roof["launches"] / steps
    sweep = sw["ms_per_frame"]
    panels = sw["panels_per_frame"]
    b_in_sweep = sw["flops_rows_of_B"] > 0
    post = sw.get("sweep_end_to_downdate_ms_per_frame", 0.0)
    gemm_ms = post * gemm_share if not b_in_sweep else 0.0              # / G
    inv_ms = post * (1.0 - gemm_share) if not b_in_sweep else 0.0       # replicated (with dx and the state update)
    st = d.get("stage_ms_per_step") or {}
    # / G: matching and RANSAC (round 6), the H P rows of the owned features and G / S by own columns (rounds 1-5: k_hp_rows,
    # k_gather_assemble -> k_g_cols + k_assemble_S on a sharded engine), from the kernel trace where there is one
    by_owner = st.get("matching_ms", 0.0) + st.get("ransac_ms", 0.0) + owner_kernels_ms
    rest = max(0.0, t1 - down - sweep - gemm_ms - inv_ms - by_owner)  # (the trace and the bench line are two runs: clamp)
    rows_m = panels * 32.0  # sum of the (padded) rows of both updates of a frame
    batches = max(1.0, (d["config"].get("mean_ransac_hypotheses") or 32.0) / 32.0)
    n_xchg = 2 + 3 + 2 * batches + 2 * 3
    print(f"{tag}: one GPU {t1:.2f} ms/frame = downdate {down:.2f} + sweep {sweep:.2f} ({panels:.0f} panels) + inverse {inv_ms:.2f} "
          f"+ GEMM {gemm_ms:.2f} + by feature ownership (matching, RANSAC, H P rows, G / S by columns) {by_owner:.2f} + replicated rest {rest:.2f}; "
          f"{n_xchg:.0f} exchanges per frame")
    for g in ranks:
        dg = down * 2.0 / g
        chain = panels * CHAIN_US * 1e-3
        sg = max(chain, sweep / g) if b_in_sweep else sweep
        gg = gemm_ms / g
        # rows of B in the sweep (<= 2048 rows): no rows of G travel (G[:, own columns] by symmetry from the own rows of P); S is
        # assembled by block columns and all-gathered: 8 m^2 bytes per update (two updates of ~62 % and ~38 % of the frame's rows)
        # (round 5: the same above 2048 rows -- the inverse + GEMM path forms G[:, own column tiles] by symmetry too; SYM_ABOVE = False
        # restores the round-4 model, where the gathered rows of G travelled there: rows_m n 8 bytes)
        recv_g = (0.53 * rows_m * rows_m * 8.0 if (b_in_sweep or SYM_ABOVE) else rows_m * n_state * 8.0) * (g - 1) / g
        recv_p = rows_m * n_state * 5.0 * (g - 1) / g
        xt = (recv_g + recv_p) / ((g - 1) * LINK_GBS * 1e9) * 1e3
        lat = n_xchg * XCHG_LAT_US * 1e-3
        tg = dg + sg + inv_ms + gg + by_owner / g + rest + xt + lat
        print(f"  G={g}: downdate {dg:.2f}  sweep {sg:.2f}  inverse {inv_ms:.2f}  GEMM {gg:.2f}  by owner {by_owner / g:.2f}  rest {rest:.2f}  "
              f"received {recv_g / 1e6:.0f} (G rows / S columns) + {recv_p / 1e6:.0f} (planes) MB -> {xt:.2f} ms + {lat:.2f} ms of exchange latency  "
              f"total {tg:.2f} ms  speed-up {t1 / tg:.2f}x")


def main():
    r = sys.argv[1] if len(sys.argv) > 1 else "r04"
    d2 = json.load(open(os.path.join(ROOT, f"{r}_bench_n2000_f32x.json")))
    own = ["k_hp_rows", "k_gather_assemble"]
    model("N=2000 (1280x720)", d2, 13 + 6 * 2000, 0.0, (2, 4, 8), per_frame_ms(os.path.join(ROOT, f"{r}_kernel_stats_n2000_f32x.csv"), own))
    d5 = json.load(open(os.path.join(ROOT, f"{r}_bench_n5000_f32x.json")))
    ks = os.path.join(ROOT, f"{r}_kernel_stats_n5000_f32x.csv")
    gemm = kernel_ms(ks, ["k_b_gemm_i8p", "k_slice_B", "k_col_exp"], 1)
    inv = kernel_ms(ks, ["k_inv_diag", "k_triinv_level", "k_dx_planes", "k_state_apply"], 1)
    share = gemm / (gemm + inv) if gemm else 0.68
    print(f"(N=5000: GEMM + its digit planes = {share:.2f} of the interval between sweep and downdate, by the kernel trace)")
    model("N=5000 (1920x1080)", d5, 13 + 6 * 5000, share, (4, 8), per_frame_ms(ks, own))


if __name__ == "__main__":
    main()
