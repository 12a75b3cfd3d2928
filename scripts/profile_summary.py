"""Turns rocprofv3 rocpd databases (kernel trace / PMC passes) into the small text summaries kept under profiles/."""
import os
import sqlite3
import sys


def kernel_stats(db_path, out_path, limit=40):
    cur = sqlite3.connect(db_path).cursor()
    rows = list(cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
    with open(out_path, "w") as f:
        f.write("kernel,calls,total_ms,avg_us,percent\n")
        for name, calls, total, avg, pct in rows[:limit]:
            f.write(f"\"{name}\",{calls},{total / 1e3:.2f},{avg:.2f},{pct:.2f}\n")


LIKE = os.environ.get("PMC_KERNEL_LIKE", "%k_p_update%")


def pmc(db_path, counter, like=LIKE):
    cur = sqlite3.connect(db_path).cursor()
    rows = list(cur.execute(
        "select kernel_name, value, duration from counters_collection where counter_name=? and kernel_name like ? "
        "order by dispatch_id", (counter, like)))
    return rows


def pmc_sq(db_paths, out_path, like=LIKE):
    """SQ / GRBM counters of the downdate kernel from several single-purpose PMC passes: mean per launch, for all
    launches and for the longer / shorter half (the low- and high-innovation updates), plus the derived ratios."""
    import statistics as st
    rows_out, means = [], {}
    for db in db_paths:
        cur = sqlite3.connect(db).cursor()
        by = {}
        for name, val, dur in cur.execute(
                "select counter_name, value, duration from counters_collection where kernel_name like ?", (like,)):
            by.setdefault(name, []).append((val, dur))
        for name, v in by.items():
            v.sort(key=lambda x: x[1])
            lo, hi = v[: len(v) // 2], v[len(v) // 2:]
            means[name] = (st.mean(x[0] for x in v), st.mean(x[0] for x in hi), st.mean(x[0] for x in lo),
                           st.mean(x[1] for x in hi) / 1e3, st.mean(x[1] for x in lo) / 1e3, len(v))
    with open(out_path, "w") as f:
        f.write(f"# {like.strip('%')}, rocprofv3 --pmc (one pass per counter group, counters only); per launch means; 'long' = the longer half of\n")
        f.write("# the launches (low-innovation updates), 'short' = the shorter half (high-innovation updates)\n")
        f.write("counter,launches,mean,mean_long,mean_short,duration_long_us,duration_short_us\n")
        for name, (m, hi, lo, dh, dl, n) in means.items():
            f.write(f"{name},{n},{m:.6g},{hi:.6g},{lo:.6g},{dh:.1f},{dl:.1f}\n")
        g = means.get("GRBM_GUI_ACTIVE")
        if g and "SQ_VALU_MFMA_BUSY_CYCLES" in means and "SQ_BUSY_CU_CYCLES" in means:
            for k, idx in (("long", 1), ("short", 2)):
                cyc = g[idx] / 8.0  # summed over the 8 XCDs
                mf = means["SQ_VALU_MFMA_BUSY_CYCLES"][idx] / 1024.0 / cyc  # 1024 SIMDs
                cu = means["SQ_BUSY_CU_CYCLES"][idx] / 256.0 / cyc
                f.write(f"# {k}: {cyc:.0f} cycles per XCD ({cyc / g[3 if idx == 1 else 4] / 1e3:.2f} GHz), MFMA pipe busy {mf:.3f} of the kernel time, "
                        f"CUs with a resident wavefront {cu:.3f}; LDS bank conflicts {means.get('SQ_LDS_BANK_CONFLICT', (0,) * 6)[idx]:.0f}\n")


def by_grid(db_path, like):
    """mean duration of a kernel's launches by grid size (the sweep: one grid size per panel index)"""
    cur = sqlite3.connect(db_path).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    gx = "grid_x" if "grid_x" in cols else ("grid_size_x" if "grid_size_x" in cols else None)
    dur = "duration" if "duration" in cols else "(end - start)"
    if gx is None:
        print("columns:", cols)
        return
    rows = list(cur.execute(f"select {gx}, count(*), avg({dur}) from kernels where name like ? group by {gx} order by {gx}", (like,)))
    for g, n, d in rows:
        print(f"grid {g:8d}  launches {n:5d}  avg {d / 1e3:8.2f} us")


if __name__ == "__main__":
    mode = sys.argv[1]
    if mode == "by_grid":
        by_grid(sys.argv[2], sys.argv[3])
        sys.exit(0)
    if mode == "pmc_sq":
        pmc_sq(sys.argv[3:], sys.argv[2])
        sys.exit(0)
    if mode == "stats":
        kernel_stats(sys.argv[2], sys.argv[3])
    elif mode == "pmc":
        # argv: fetch.db write.db out.csv [launches to skip: the warm-up frames' two launches each + set_state's first]
        skip = int(sys.argv[5]) if len(sys.argv) > 5 else 0
        fetch = pmc(sys.argv[2], "FETCH_SIZE")[skip:]
        write = pmc(sys.argv[3], "WRITE_SIZE")[skip:]
        with open(sys.argv[4], "w") as f:
            f.write("# HBM traffic of the downdate kernel per launch (rocprofv3 --pmc, separate passes; values in KB as reported),\n")
            f.write(f"# bench.py's own timed frames (the first {skip} launches = warm-up frames dropped)\n")
            f.write("# FETCH_SIZE on gfx950 counts wide coalesced reads at 1/2 (MI355X_MICROARCH.md, HBM section): x2 below\n")
            f.write("launch,fetch_size_kb_raw,fetch_bytes_corrected,write_size_kb,write_bytes,duration_us\n")
            for i, (fr, wr) in enumerate(zip(fetch, write)):
                f.write(f"{i},{fr[1]:.1f},{fr[1] * 1024 * 2:.0f},{wr[1]:.1f},{wr[1] * 1024:.0f},{fr[2] / 1e3:.1f}\n")
            if fetch and write:
                mf = sum(r[1] for r in fetch) / len(fetch) * 1024 * 2
                mw = sum(r[1] for r in write) / len(write) * 1024
                f.write(f"# mean per launch: fetch {mf / 1e6:.1f} MB (corrected), write {mw / 1e6:.1f} MB, total {(mf + mw) / 1e6:.1f} MB\n")
                # per launch class: the low- and high-innovation updates alternate; the longer launch of a pair is the m >= 512 one
                cut = sorted(r[2] for r in fetch)[len(fetch) // 2]
                for name, sel in (("m>=512", lambda d: d >= cut), ("m<512", lambda d: d < cut)):
                    fs = [r[1] for r in fetch if sel(r[2])]
                    ws = [r[1] for r in write if sel(r[2])]
                    ds = [r[2] for r in fetch if sel(r[2])]
                    if fs and ws:
                        cf, cw = sum(fs) / len(fs) * 2048, sum(ws) / len(ws) * 1024
                        f.write(f"# class {name}: launches {len(fs)} mean {sum(ds) / len(ds) / 1e3:.1f} us fetch {cf / 1e6:.1f} MB write {cw / 1e6:.1f} MB total {(cf + cw) / 1e6:.1f} MB\n")
