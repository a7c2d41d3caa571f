"""HBM traffic per launch of every MFMA kernel instantiation from two rocprofv3 PMC passes of bench.py.

    python tools/make_traffic.py <tag>_FETCH_SIZE_pmc.csv <tag>_WRITE_SIZE_pmc.csv profiles/traffic.json [config] [batch]

`config` (c4 / c2 / c3 / sams, default c4) selects the section of traffic.json that is replaced; bench.py reads the section
of the configuration it runs.

The passes are collected by tools/profile_round.sh (`--pmc FETCH_SIZE` and `--pmc WRITE_SIZE` in separate runs with
--kernel-trace only).  Units and correction follow /opt/skills/guides/MI355X_MICROARCH.md §HBM: both counters are in
KiB; on gfx950 FETCH_SIZE reports half the bytes of wide coalesced reads (TCC_EA0_RDREQ tallied at 64 B for 128-B
requests), so it is doubled.  WRITE_SIZE was checked against a known byte count here: a split-K=2 launch writing two
[24576 x 256] fp32 slabs reports exactly 49152 KiB.  bench.py copies `hbm_bytes_per_launch` of the dominant
instantiation into `roofline.traffic`."""
import csv
import json
import re
import sys

MODES = {0: "fprop", 1: "dgrad", 2: "wgrad", 3: "gemm"}


def key_of(sym):
    if "wino_fused_k" in sym:
        return "winograd_fused"
    if "pgemm_nt_k" in sym:
        return "winograd_pgemm"
    m = re.match(r"_Z15so_igemm_kernelILi(\d)ELb[01]ELb[01]ELi(\d+)ELi(\d+)ELi(\d)EE", sym)
    if not m:
        return None
    mode, bm, bn, nw = int(m.group(1)), int(m.group(2)), int(m.group(3)), int(m.group(4))
    return f"{MODES[mode]}_{bm}x{bn}" + ("w8" if nw == 8 else "")


def load(path, col):
    out = {}
    for r in csv.DictReader(open(path)):
        k = key_of(r["Name"])
        if k and r.get(col):
            a = out.setdefault(k, [0, 0.0])
            a[0] += int(r["Calls"])
            a[1] += float(r[col])
    return out


def main(fetch_csv, write_csv, out_json, config="c4", batch=None):
    import os

    f, w = load(fetch_csv, "FETCH_SIZE"), load(write_csv, "WRITE_SIZE")
    allcfg = json.load(open(out_json)) if os.path.exists(out_json) else {}
    if "_doc" in allcfg and "c4" not in allcfg:   # round-2 layout (one flat section = c4): start over
        allcfg = {}
    allcfg["_doc"] = ("HBM-side bytes per launch from rocprofv3 PMC (FETCH_SIZE x2 gfx950 correction, WRITE_SIZE), KiB -> bytes; "
                      "one section per bench.py --config")
    # `_batch`: frames per GPU of the profiled command; bench.py reports `roofline.traffic` only for a run at that batch
    res = {"_sources": [fetch_csv.split("/")[-1], write_csv.split("/")[-1]],
           "_batch": int(batch) if batch is not None else (2 if config == "c5" else 4)}
    for k in sorted(f):
        fk = f[k][1] / f[k][0]
        wk = w[k][1] / w[k][0] if k in w else 0.0
        res[k] = {"launches_sampled": f[k][0], "fetch_kib_per_launch_raw": round(fk, 1), "write_kib_per_launch": round(wk, 1),
                  "hbm_bytes_per_launch": int((2.0 * fk + wk) * 1024)}
    allcfg[config] = res
    json.dump(allcfg, open(out_json, "w"), indent=1)
    print(f"{len(res) - 2} instantiations -> {out_json} [{config}]")


if __name__ == "__main__":
    main(*sys.argv[1:6])
