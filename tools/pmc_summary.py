#!/usr/bin/env python3
"""Summarise two rocprofv3 counter passes (FETCH_SIZE, WRITE_SIZE; collected separately, --pmc + --kernel-trace only)
into profiles/rNN_pmc_gemm.json (pass the round's name as the 4th argument) for one kernel-name substring (or several, `a|b`: averaged over
the launches of all of them — the 8-phase GEMM runs as gemm_nt_bf16_8phase_kernel and gemm_nt_bf16_tall_kernel).

  python tools/pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> [kernel substring] [out.json]

Corrections follow MI355X_MICROARCH.md (HBM section): FETCH_SIZE is in KB and on gfx950 reports half the bytes of wide
coalesced reads (128-B requests tallied at 64 B) -> doubled; WRITE_SIZE (KB) is exact for 16-B-per-lane stores."""
import csv, json, sys


def avg_counter(path, counter, sub):
    tot, n = 0.0, 0
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter and any(x in r["Kernel_Name"] for x in sub.split("|")):
            tot += float(r["Counter_Value"])
            n += 1
    return (tot / n if n else 0.0), n


def main():
    f, w = sys.argv[1], sys.argv[2]
    sub = sys.argv[3] if len(sys.argv) > 3 else "gemm_nt_bf16_8phase_kernel"
    out = sys.argv[4] if len(sys.argv) > 4 else "profiles/rNN_pmc_gemm.json"       # bench.py reads the newest profiles/r[0-9][0-9]_pmc_gemm.json
    fk, n1 = avg_counter(f, "FETCH_SIZE", sub)
    wk, n2 = avg_counter(w, "WRITE_SIZE", sub)
    rec = {
        "kernel": sub + " (all template instances)",
        "launches": n1,
        "FETCH_SIZE_avg_KB": fk,
        "WRITE_SIZE_avg_KB": wk,
        "hbm_bytes_per_launch": 2.0 * fk * 1024.0 + wk * 1024.0,
        "note": "separate rocprofv3 --pmc passes (FETCH_SIZE, then WRITE_SIZE) over bench.py --steps 1 --warmup 1; FETCH_SIZE doubled "
                "per MI355X_MICROARCH.md HBM section (gfx950 tallies 128-B requests at 64 B); average over every launch of the kernel "
                "(warm-up + timed step); counts L2<->fabric requests, Infinity-Cache hits included",
    }
    json.dump(rec, open(out, "w"), indent=1)
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
