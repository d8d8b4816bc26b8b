#!/usr/bin/env python3
"""Per-launch PMC counters of every kernel of one frame of the default bench workload -> JSON.

    python3 tools/collect_pmc.py --commit <hash> --out gpurun_out/r03_pmc.json        (on the GPU box)

One rocprofv3 pass per counter group (FETCH_SIZE and WRITE_SIZE cannot share a pass on gfx950), each over
`bench.py --steps 3 --warmup 1 --graph off --inflight 1` (eager launches, one frame in flight: the counters of a launch
are then that launch's alone).  This script never touches the GPU itself; rocprofv3 starts bench.py directly.
FETCH_SIZE / WRITE_SIZE are in KiB as rocprofv3 reports them; bench.py applies the guide's x2 on the read side.
The file is copied to profiles/rNN_pmc.json and committed; bench.py quotes it with the commit and library hash."""
import argparse
import collections
import csv
import glob
import hashlib
import json
import os
import shutil
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GROUPS = ["FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH",
          "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES",
          "TCC_HIT_sum TCC_MISS_sum"]
NAMES = {"k_render_binned": "k_render_binned", "k_bin_count": "k_bin_count", "k_prep": "k_prep",
         "k_zero_counters": "k_zero_counters"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--commit", required=True)
    ap.add_argument("--out", required=True)
    args = ap.parse_args()
    lib = os.environ.get("SRH_LIB") or os.path.join(REPO, "surf_renderer_amd", "libsrh.so")
    sha = hashlib.sha256(open(lib, "rb").read()).hexdigest()
    kernels = collections.defaultdict(dict)
    env = dict(os.environ, TMPDIR="/tmp")
    for group in GROUPS:
        tmp = os.path.join(REPO, "gpurun_out", "pmc_tmp")
        shutil.rmtree(tmp, ignore_errors=True)
        cmd = ["rocprofv3", "--pmc", *group.split(), "--output-format", "csv", "-d", tmp, "--", sys.executable,
               os.path.join(REPO, "bench.py"), "--no-cpu-baseline", "--no-check", "--warmup-ms", "0", "--steps", "3",
               "--warmup", "1", "--graph", "off", "--inflight", "1"]
        proc = subprocess.run(cmd, cwd=REPO, env=env, capture_output=True, text=True, timeout=400)
        files = glob.glob(os.path.join(tmp, "*", "*counter_collection.csv"))
        if proc.returncode != 0 or not files:
            print(f"[pmc] pass '{group}' failed ({proc.returncode}): {proc.stderr[-400:]}", file=sys.stderr)
            continue
        acc = collections.defaultdict(lambda: collections.defaultdict(float))
        cnt = collections.defaultdict(collections.Counter)
        for row in csv.DictReader(open(files[0])):
            key = next((v for k, v in NAMES.items() if k in row["Kernel_Name"]), None)
            if key is None:
                continue
            acc[key][row["Counter_Name"]] += float(row["Counter_Value"])
            cnt[key][row["Counter_Name"]] += 1
        for key in acc:
            for c, v in acc[key].items():
                kernels[key][c] = v / cnt[key][c]
            kernels[key]["launches_averaged"] = max(cnt[key].values())
        print(f"[pmc] {group}: ok", flush=True)
    out = {"commit": args.commit, "lib_sha256": sha,
           "workload": "bench.py default: 100k discs, 2048x2048, 4 lights, binned; --graph off --inflight 1 --steps 3",
           "units": "FETCH_SIZE / WRITE_SIZE in KiB per launch, the rest in counts per launch (SQ_*_CYCLES in quad-cycles)",
           "kernels": kernels}
    with open(args.out, "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
    print(json.dumps(kernels.get("k_render_binned", {})))


if __name__ == "__main__":
    main()
