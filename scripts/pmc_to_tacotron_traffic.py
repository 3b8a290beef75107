#!/usr/bin/env python
"""HBM bytes of one Tacotron configs[2] pass from the FETCH_SIZE / WRITE_SIZE passes of scripts/profile_secondaries.sh ->
gpurun_out/prof_<tag>/summary_tacotron_traffic_<tag>.txt and profiles/traffic.json["tacotron:" + _lib.tacotron_hash()].

FETCH_SIZE / WRITE_SIZE are in KiB; per MI355X_MICROARCH.md (HBM section) FETCH_SIZE on gfx950 counts 64 B per 128-B request of a wide
coalesced read -> doubled (the same correction as scripts/pmc_to_traffic.py); WRITE_SIZE is taken as reported.  The profiled command
runs 4 passes (one warm-up + 3): per-pass figures = sums over the launches / number of tc_decoder_g_kernel launches."""
import csv, glob, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
out, tag = sys.argv[1], sys.argv[2]


def rows(pattern):
    for f in glob.glob(os.path.join(out, pattern), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                yield r


def short(name):
    m = re.search(r"(tc_\w+)", name)
    return m.group(1) if m else None


def collect(prefix):
    agg = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        for r in rows("%s_%s/**/*counter_collection.csv" % (prefix, c)):
            k = short(r["Kernel_Name"])
            if k is None:
                continue
            a = agg.setdefault(k, {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0, "n": {"FETCH_SIZE": set(), "WRITE_SIZE": set()}})
            a[r["Counter_Name"]] += float(r["Counter_Value"])
            a["n"][r["Counter_Name"]].add(r["Dispatch_Id"])
    return agg


def table(agg, title):
    dec = [k for k in agg if k.startswith("tc_decoder")]
    passes = max(1, max([len(agg[k]["n"]["FETCH_SIZE"]) for k in dec] or [1]))
    lines = [title + " -- HBM traffic per kernel and pass from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate runs, %d passes each; FETCH_SIZE KiB x 2: "
             "gfx950 correction)" % passes]
    tot = {"gemm_fetch": 0.0, "gemm_write": 0.0, "fetch": 0.0, "write": 0.0, "dec_fetch": 0.0, "dec_write": 0.0, "dec_kernel": None}
    for k, a in sorted(agg.items(), key=lambda kv: -(kv[1]["FETCH_SIZE"] * 2 + kv[1]["WRITE_SIZE"])):
        f = 2.0 * 1024.0 * a["FETCH_SIZE"] / passes
        w = 1024.0 * a["WRITE_SIZE"] / passes
        lines.append("  %-36s launches/pass %4d   read %10.2f MB   written %10.2f MB" % (k, len(a["n"]["FETCH_SIZE"]) // passes, f / 1e6, w / 1e6))
        tot["fetch"] += f; tot["write"] += w
        if k.startswith("tc_gemm"):
            tot["gemm_fetch"] += f; tot["gemm_write"] += w
        if k.startswith("tc_decoder"):
            tot["dec_fetch"] += f; tot["dec_write"] += w; tot["dec_kernel"] = k
    lines.append("  => the matrix-core kernels (tc_gemm_*): %.1f MB read + %.1f MB written per pass; the decoder (%s): %.1f MB read + %.1f MB written; "
                 "the whole pass: %.1f MB read + %.1f MB written"
                 % (tot["gemm_fetch"] / 1e6, tot["gemm_write"] / 1e6, tot["dec_kernel"], tot["dec_fetch"] / 1e6, tot["dec_write"] / 1e6, tot["fetch"] / 1e6, tot["write"] / 1e6))
    return lines, tot


import twvk_amd
h = "tacotron:" + twvk_amd._lib.tacotron_hash()
p = os.path.join(ROOT, "profiles", "traffic.json")
t = json.load(open(p)) if os.path.exists(p) else {}
all_lines = []
# pmc_taco: the default kernels at B = 32 (the XCD-resident decoder); pmc_tacog32: B = 32 on the split decoder (decoder_groups = 8);
# pmc_taco16: B = 16
for prefix, key, title in (("pmc_taco", "B32_T101", "Tacotron configs[2] pass (B = 32, 101 tokens, 200 decoder steps)"),
                           ("pmc_tacog32", "B32_T101_split", "the same pass with the split decoder forced (decoder_groups = 8: the default until round 6)"),
                           ("pmc_taco16", "B16_T101", "B = 16")):
    agg = collect(prefix)
    if not agg:
        continue
    lines, tot = table(agg, title)
    all_lines += lines + [""]
    t.setdefault(h, {})[key] = {"gemm_fetch_bytes_per_pass": tot["gemm_fetch"], "gemm_write_bytes_per_pass": tot["gemm_write"],
                                "decoder_kernel": tot["dec_kernel"], "decoder_fetch_bytes_per_pass": tot["dec_fetch"], "decoder_write_bytes_per_pass": tot["dec_write"],
                                "pass_fetch_bytes": tot["fetch"], "pass_write_bytes": tot["write"],
                                "profile": "profiles/%s_rocprofv3_tacotron_traffic.txt" % tag}
if not all_lines:
    sys.exit("no Tacotron FETCH_SIZE / WRITE_SIZE passes under " + out)
json.dump(t, open(p, "w"), indent=1, sort_keys=True)
all_lines.append("(profiles/traffic.json updated under %s)" % h)
open(os.path.join(out, "summary_tacotron_traffic_%s.txt" % tag), "w").write("\n".join(all_lines) + "\n")
print("\n".join(all_lines))
