"""profiles/traffic.json from the FETCH_SIZE / WRITE_SIZE PMC passes of bench.py (one pass per counter, nothing else traced):
HBM bytes of ONE launch of the count kernel, keyed by '<reads>_k<K>_<mode>' -- what bench.py reports as roofline.traffic.
usage: python tools/make_traffic.py <key> <fetch counter_collection.csv> <write counter_collection.csv> [...more triples]
gfx950 correction (MI355X_MICROARCH.md, HBM/rocprofv3): FETCH_SIZE counts the 128-B requests of a wide coalesced stream
at 64 B -> x2 for the record stream the count kernel reads; WRITE_SIZE as reported (KB)."""
import csv, json, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
out = ROOT / "profiles" / "traffic.json"
doc = json.loads(out.read_text()) if out.exists() else {}
entries = doc.get("entries", {})


def per_launch(path, counter, kernel="snk_count_kernel"):
    """Average counter value of the kernel's FULL launches (a first call also has the short pilot launch over 1/64 of the buckets and
    the launch that finishes it: anything under a quarter of the largest value is left out, the finishing launch counts as full)."""
    vals, name = [], None
    for r in csv.DictReader(open(path)):
        if kernel in r["Kernel_Name"] and r["Counter_Name"] == counter:
            vals.append(float(r["Counter_Value"])); name = r["Kernel_Name"].split("(")[1 if r["Kernel_Name"].startswith("void (") else 0]
    if not vals:
        return 0.0, 0, name
    mx = max(vals)
    full = [v for v in vals if v >= 0.9 * mx] or vals
    return sum(full) / len(full), len(full), name


a = sys.argv[1:]
if a and a[0] == "--instmix":
    # python tools/make_traffic.py --instmix <key> <SQ counter_collection.csv>: wave-instruction counts per full launch of the two big
    # kernels (SQ_INSTS_VALU / _SALU / _LDS ..., SQ_WAVE_CYCLES, SQ_BUSY_CYCLES as collected by tools/prof_round.sh) -- what bench.py
    # prices the count kernel's VALU-issue roofline with
    key, path = a[1], a[2]
    names = sorted({r["Counter_Name"] for r in csv.DictReader(open(path))})
    e = entries.setdefault(key, {})
    for kern, tag in (("snk_count_kernel", "count_kernel"), ("snk_msp_kernel", "partition_kernel")):
        for cn in names:
            v, n, _ = per_launch(path, cn, kern)
            if n:
                e[f"{tag}_{cn}_per_launch"] = v
    e["instmix_source"] = Path(path).name
    print(key, {k: v for k, v in e.items() if "SQ_" in k or "GRBM" in k})
    doc["entries"] = entries
    out.write_text(json.dumps(doc, indent=1) + "\n")
    sys.exit(0)
for i in range(0, len(a), 3):
    key, f_csv, w_csv = a[i:i + 3]
    f, nf, name = per_launch(f_csv, "FETCH_SIZE")
    w, nw, _ = per_launch(w_csv, "WRITE_SIZE")
    pf, _, _ = per_launch(f_csv, "FETCH_SIZE", "snk_msp_kernel")
    pw, _, _ = per_launch(w_csv, "WRITE_SIZE", "snk_msp_kernel")
    keep = {k: v for k, v in entries.get(key, {}).items() if "SQ_" in k or "GRBM" in k or k == "instmix_source"}
    entries[key] = {**keep, "FETCH_SIZE_KB_per_launch": f, "WRITE_SIZE_KB_per_launch": w, "launches_averaged": [nf, nw],
                    "count_kernel_hbm_bytes_per_launch": (2 * f + w) * 1024,
                    # the minimiser partition: packed rows streamed in (x2 as above), 32-byte records scattered out (WRITE_SIZE counts the
                    # half-filled 64-byte sectors in full: 2.1x the payload)
                    "partition_FETCH_SIZE_KB_per_launch": pf, "partition_WRITE_SIZE_KB_per_launch": pw,
                    "partition_kernel_hbm_bytes_per_launch": (2 * pf + pw) * 1024,
                    "source": [Path(f_csv).name, Path(w_csv).name]}
    print(key, entries[key])
doc = {"note": "HBM bytes per launch of snk_count_kernel = (2 x FETCH_SIZE + WRITE_SIZE) KB from separate rocprofv3 --pmc passes of "
               "bench.py (--steps 1 --warmup 1); FETCH_SIZE x2 = the gfx950 correction for wide coalesced reads (MI355X_MICROARCH.md)",
       "entries": entries}
out.write_text(json.dumps(doc, indent=1) + "\n")
