#!/usr/bin/env python3
"""Per-launch PMC counters of the shadow pass and the backward kernels (tools/prof_bwd_shadow.py cases) -> JSON.
Same recipe as tools/collect_pmc.py: one rocprofv3 pass per counter group, counters averaged per kernel name.

    python3 tools/collect_pmc_aux.py --commit <hash> --out gpurun_out/r03_pmc_aux.json       (on the GPU box)
"""
import argparse, collections, csv, glob, hashlib, json, os, shutil, subprocess, sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GROUPS = ["FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH",
          "SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES", "TCC_HIT_sum TCC_MISS_sum"]
KERNELS = ("k_shadow_shade_binned", "k_prep_views", "k_scene_bounds", "k_render_bwd_tch", "k_render_bwd(")


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
        tmp = os.path.join(REPO, "gpurun_out", "pmc_tmp_aux")
        shutil.rmtree(tmp, ignore_errors=True)
        cmd = ["rocprofv3", "--pmc", *group.split(), "--output-format", "csv", "-d", tmp, "--", sys.executable,
               os.path.join(REPO, "tools", "prof_bwd_shadow.py"), "--cases", "shadow_cfg5,bwd_mesh_resident,bwd_mesh_resident_tch",
               "--steps", "2"]
        proc = subprocess.run(cmd, cwd=REPO, env=env, capture_output=True, text=True, timeout=500)
        files = glob.glob(os.path.join(tmp, "*", "*counter_collection.csv"))
        if proc.returncode != 0 or not files:
            print(f"[pmc] pass '{group}' failed ({proc.returncode}): {proc.stderr[-400:]}", file=sys.stderr)
            continue
        acc = collections.defaultdict(lambda: collections.defaultdict(float))
        cnt = collections.defaultdict(collections.Counter)
        mx = collections.defaultdict(lambda: collections.defaultdict(float))
        for row in csv.DictReader(open(files[0])):
            key = next((k.rstrip("(") for k in KERNELS if k in row["Kernel_Name"]), None)
            if key is None:
                continue
            v = float(row["Counter_Value"])
            acc[key][row["Counter_Name"]] += v
            cnt[key][row["Counter_Name"]] += 1
            mx[key][row["Counter_Name"]] = max(mx[key][row["Counter_Name"]], v)
        for key in acc:
            for c, v in acc[key].items():
                kernels[key][c] = v / cnt[key][c]
                kernels[key][c + "_max"] = mx[key][c]
            kernels[key]["launches_averaged"] = max(cnt[key].values())
        print(f"[pmc] {group}: ok", flush=True)
    out = {"commit": args.commit, "lib_sha256": sha,
           "workload": "tools/prof_bwd_shadow.py --cases shadow_cfg5,bwd_mesh_resident,bwd_mesh_resident_tch (the shadow "
                       "kernel's *_max entries are the 2048^2 x 100 k-disc launches; its averages include warm-up frames of other sizes)",
           "units": "FETCH_SIZE / WRITE_SIZE in KiB per launch, the rest in counts per launch",
           "kernels": kernels}
    with open(args.out, "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
    print(json.dumps({k: {c: round(v) for c, v in d.items() if c.startswith("SQ_INSTS_VALU") or c.startswith("FETCH") or c.startswith("WRITE")} for k, d in kernels.items()}))


if __name__ == "__main__":
    main()
