#!/usr/bin/env python
"""profiles/pmc_traffic.json <- the FETCH_SIZE / WRITE_SIZE passes of tools/gpu_round6_evidence.sh (CPU; run in the builder's
tree after gpurun merged gpurun_out/ back).  Every node-launch entry names the dd_attention2.hip it was measured on
(kernel_source_sha256_16, written by the evidence script on the GPU box) and the commit: bench.py refuses an entry whose hash
is not the current source's.   usage: python tools/pmc_traffic_update.py gpurun_out/r4fin <commit> profiles/round4_pmc_node"""
import json, os, re, sys
out_dir, commit, src_note = sys.argv[1], sys.argv[2], sys.argv[3]
sha = open(os.path.join(out_dir, "kernel_source_sha256_16.txt")).read().strip()


def node_row(path):
    best = None
    for line in open(path):
        if "k_attn2_node" not in line:
            continue
        c = [x.strip() for x in line.strip().strip("|").split("|")]
        calls, avg, wgs = int(c[3]), float(c[4]), int(c[1])
        if best is None or calls > best[0]:
            best = (calls, avg, wgs, c[0])
    return best


def pair(f_md, w_md, key, label):
    f, w = node_row(os.path.join(out_dir, f_md)), node_row(os.path.join(out_dir, w_md))
    return key, {"kernel": f"{label} ({f[2]} workgroups in the FETCH pass, {w[2]} in the WRITE pass: the CU split is measured per process)",
                 "fetch_kib_per_launch": round(f[1], 1), "write_kib_per_launch": round(w[1], 1),
                 "kernel_source_sha256_16": sha, "measured_at_commit": commit, "source": src_note,
                 "note": f"FETCH_SIZE + WRITE_SIZE per launch, rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes "
                         f"(tools/gpu_round6_evidence.sh), averages over the {f[0]} / {w[0]} launches of the grid each process settled "
                         "on; KiB as reported; the gfx950 FETCH_SIZE under-count of wide coalesced reads (guide: up to 2x) is not "
                         "applied because the kernel mixes 16-byte and 4-byte gathers"}


path = "profiles/pmc_traffic.json"
pm = json.load(open(path))
for args in (("pmc_small_3.md", "pmc_small_4.md", "NP300_NL30_B8", "dd::v2::k_attn2_node<2,8>"),
             ("pmc_large_5.md", "pmc_large_6.md", "NP600_NL60_B8", "dd::v2::k_attn2_node<4,8>")):
    try:
        k, ent = pair(*args)
        pm["node_launch"][k] = ent
        print(k, ent["fetch_kib_per_launch"], ent["write_kib_per_launch"], sha)
    except Exception as e:                                   # noqa: BLE001
        print("skipped", args[2], e)
json.dump(pm, open(path, "w"), indent=1)
