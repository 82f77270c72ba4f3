#!/usr/bin/env python
"""A/B of builds that also need their own environment (e.g. a packing switch on the Python side).
usage: python tools/ab_env.py [rounds] name=lib.so[,ENV=VAL...] name2=lib2.so[,ENV=VAL...]
Per variant: SHA-256 of seeded chains (tools/lib_checksum.py) once, then alternating timing runs (the child of
tools/ab_builds.py).  DD_WORKLOAD / DD_B as there."""
import os, re, statistics, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from ab_builds import CHILD
args = sys.argv[1:]
rounds = int(args.pop(0)) if args and args[0].isdigit() else 3
variants = []
for a in args:
    name, rest = a.split("=", 1)
    parts = rest.split(",")
    env = dict(kv.split("=", 1) for kv in parts[1:])
    variants.append((name, os.path.abspath(parts[0]), env))
if os.environ.get("AB_CHECKSUM", "1") == "1":
    for name, lib, env in variants:
        out = subprocess.run([sys.executable, "tools/lib_checksum.py"], env=dict(os.environ, DD_HIP_LIB=lib, DD_IGNORE_ABI="1", **env),
                             capture_output=True, text=True)
        lines = [l for l in out.stdout.splitlines() if re.match(r"^(small|mid37|large) ", l)]
        print(f"{name:12s} checksums: {'  '.join(lines) if lines else 'FAILED ' + out.stderr[-600:]}", flush=True)
res = {v[0]: [] for v in variants}
for r in range(rounds):
    for name, lib, env in variants:
        out = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, DD_HIP_LIB=lib, DD_IGNORE_ABI="1", **env),
                             capture_output=True, text=True)
        if out.returncode != 0:
            print(name, "FAILED", out.stderr[-800:]); continue
        res[name].append(float(out.stdout.strip().splitlines()[-1]))
for name, _, _ in variants:
    if res[name]:
        print(f"{name:12s} median {statistics.median(res[name]):.4f} ms/step  all {[round(x, 4) for x in res[name]]}", flush=True)
