"""Turns the two rocprofv3 --pmc passes of profiles/pmc_traffic.sh (FETCH_SIZE, WRITE_SIZE; separate passes, no tracing)
into the per-kernel HBM-traffic table committed as profiles/<tag>_pmc_hbm_traffic.txt and the dominant kernel's row
profiles/pmc_traffic.json that bench.py copies into roofline.traffic.

  python profiles/pmc_summarize.py <tag> [<dominant kernel name> [<json file name> [<bench arguments, for the header>]]]

Units and corrections (MI355X_MICROARCH.md, HBM section): the counters are KiB as rocprofv3 reports them; on gfx950
FETCH_SIZE tallies wide (16 B per lane) streaming reads at HALF their bytes, WRITE_SIZE is exact, so the HBM-side bytes
of a launch are ~ 2 * FETCH_SIZE + WRITE_SIZE.  Both raw columns are kept so the correction can be undone."""
import collections
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load(counter):
    path = os.path.join(ROOT, "gpurun_out", "pmc_traffic_%s" % counter, "p_counter_collection.csv")
    tot, n = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"]
        if k.startswith("void "):
            k = k[5:]
        k = k.split("(")[0]
        tot[k] += float(r["Counter_Value"])
        n[k] += 1
    return tot, n


TRAFFIC_SOURCES = ("jafpro_amd/csrc/conv_dma.hip", "jafpro_amd/csrc/conv_dma_kernel.h", "jafpro_amd/ops.py", "jafpro_amd/crn_model.py",
                   "jafpro_amd/networks.py")


def git_blob_hash(path):
    """`git hash-object <path>` without git (sha1 of "blob <size>\0" + contents)."""
    import hashlib
    data = open(path, "rb").read()
    return hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()


def source_blobs():
    return {p: git_blob_hash(os.path.join(ROOT, p)) for p in TRAFFIC_SOURCES}


def main():
    tag = sys.argv[1]
    fetch, nf = load("FETCH_SIZE")
    write, nw = load("WRITE_SIZE")
    rows = []
    for k in fetch:
        if k not in write or nf[k] != nw[k]:
            continue
        f, w = fetch[k] / nf[k], write[k] / nw[k]
        rows.append((k, nf[k], f, w, (2 * f + w) * 1024 / 1e6, (2 * fetch[k] + write[k]) * 1024))
    rows.sort(key=lambda r: -r[5])
    out = os.path.join(ROOT, "profiles", "%s_pmc_hbm_traffic.txt" % tag)
    with open(out, "w") as fh:
        extra = (" " + sys.argv[4]) if len(sys.argv) > 4 and sys.argv[4] else ""
        fh.write("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, profiles/pmc_traffic.sh) over\n"
                 "# `bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-config2 --parity-mode-steps 0 --no-frame-parity --serial-streams" + extra + "`\n"
                 "# (3 train steps, bf16 packed path).  KiB as rocprofv3 reports them; gfx950: wide streaming reads are tallied at half\n"
                 "# their bytes, writes exactly (MI355X_MICROARCH.md, HBM section): est. HBM MB per launch = (2*FETCH + WRITE) KiB.\n"
                 "# Rows sorted by total estimated bytes; every kernel of the step is listed (the gather / blend kernels of the\n"
                 "# north star -- texture_warp, flow_warp, blend, bc_transform, grid_sample -- are the short rows near the end).\n")
        fh.write("%-56s %8s %16s %16s %18s\n" % ("kernel", "launches", "FETCH_KiB/launch", "WRITE_KiB/launch", "est_HBM_MB/launch"))
        for k, n, f, w, mb, _ in rows:
            fh.write("%-56s %8d %16.1f %16.1f %18.2f\n" % (k[:56], n, f, w, mb))
    dom = sys.argv[2] if len(sys.argv) > 2 and sys.argv[2] else "conv_dma_kernel<4, 4, false, false, false>"
    jname = sys.argv[3] if len(sys.argv) > 3 and sys.argv[3] else "pmc_traffic.json"
    for k, n, f, w, mb, _ in rows:
        if k == dom:
            json.dump({"kernel": dom, "launches": n, "fetch_kib_per_launch": f, "write_kib_per_launch": w,
                       "hbm_bytes_per_launch": (2 * f + w) * 1024,
                       # provenance: the commit the tree was at (JAF_PROFILE_COMMIT, passed by the caller of the
                       # profiling script: the GPU box has no .git) and the git blob hashes of the sources that shape the
                       # dominant kernel's traffic; bench.py recomputes the hashes and flags a stale figure
                       "commit": os.environ.get("JAF_PROFILE_COMMIT"), "kernel_source_blobs": source_blobs(),
                       "source": "profiles/%s_pmc_hbm_traffic.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; "
                                 "2*FETCH+WRITE per MI355X_MICROARCH.md)" % tag},
                      open(os.path.join(ROOT, "profiles", jname), "w"), indent=1)
            print("dominant", dom, "%.1f MB per launch" % ((2 * f + w) * 1024 / 1e6))
    print("wrote", out, len(rows), "kernels")


if __name__ == "__main__":
    main()
