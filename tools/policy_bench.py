#!/usr/bin/env python3
"""Throughput of the reference's REAL workload on one MI355X: the sender's policy sweep (city_sender.py:495-607) -- per
video 2 quality indexes x 28 thresholds = 56 (video, q, threshold) jobs, each a receiver loop of key frames and generated
chunks -- through ``evc_amd.policy.run_policy``, which advances all jobs in lockstep and stacks them along the batch axis
of the score-network launches.  The reference runs the 56 jobs one after another (and reloads the 1 GB checkpoint per
chunk).  Synthetic clips, seeded weights of the reference architecture, the PSNR rule (LPIPS needs AlexNet weights that
are not available offline); thresholds are spread over the PSNR range the generated frames actually reach, so that the
sweep contains both accepted chunks and key-frame fall-backs like a real one.

    python tools/policy_bench.py [--videos 1] [--subsample 100] [--max-batch 32] > profiles/r03_policy_bench.json
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import evc_amd  # noqa: E402,F401
from evc_amd import lib as L, policy as P, sampler as S, synthetic  # noqa: E402
from evc_amd.config import default_config  # noqa: E402
from evc_amd.decoder import ClipDecoder  # noqa: E402
from evc_amd.elic import ElicModel  # noqa: E402
from evc_amd.scorenet import ScoreNet  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--videos", type=int, default=1)
    ap.add_argument("--subsample", type=int, default=100)
    ap.add_argument("--max-batch", type=int, default=32)
    ap.add_argument("--qs", type=int, nargs="+", default=[4, 5])             # city_sender.py:504 q range of the sweep
    ap.add_argument("--n-thresholds", type=int, default=28)                  # city_sender.py:505-508
    a = ap.parse_args()
    L.hip_lib()
    cfg = default_config(192, 192, 128, subsample=a.subsample)
    net = ScoreNet(cfg, synthetic.diffusion_state_dict(cfg, 1234))
    models = {q: ElicModel(synthetic.elic_state_dict(q)) for q in a.qs}
    dec = ClipDecoder(net, models[a.qs[0]], cfg, S.get_sampler("DDPM"))
    clips = {v: torch.from_numpy(synthetic.make_clips(1, seed=200 + v)[0].astype(np.float32) / 255.0) for v in range(a.videos)}
    # probe: one all-accepting job at 10 sampler steps to find the PSNR range generated frames reach on this data
    probe_cfg = default_config(192, 192, 128, subsample=10)
    pdec = ClipDecoder(net, models[a.qs[0]], probe_cfg, S.get_sampler("DDPM"))
    r = P.run_policy(pdec, models, {0: clips[0]}, a.qs[:1], [-1e9], P.PsnrMetric(), max_batch=a.max_batch)
    x = r[(0, a.qs[0])][0]["x"]
    ps = [P.cal_psnr(x[f], clips[0][f].numpy()) for f in range(2, 30)]
    lo, hi = float(np.min(ps)) - 1.0, float(np.max(ps)) + 1.0
    thresholds = [float(t) for t in np.linspace(lo, hi, a.n_thresholds)]
    torch.cuda.synchronize()
    stats = {}
    t0 = time.perf_counter()
    res = P.run_policy(dec, models, clips, a.qs, thresholds, P.PsnrMetric(), max_batch=a.max_batch, stats=stats,
                       log=lambda m: print(f"[policy_bench {time.strftime('%H:%M:%S')}] {m}", file=sys.stderr, flush=True))
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    jobs = a.videos * len(a.qs) * len(thresholds)
    kept = sum(len(v) for v in res.values())
    gen = sum(int((r_["d"] == 0).sum()) for v in res.values() for r_ in v)
    launches = sum(stats["launch_sizes"].values())
    fwd = launches * (a.subsample + 1)
    samples = sum(k * n for k, n in stats["launch_sizes"].items()) * (a.subsample + 1)
    print(json.dumps({
        "workload": f"{a.videos} video(s) x q {a.qs} x {len(thresholds)} PSNR thresholds = {jobs} sender jobs of 30 frames "
                    f"(city_sender.py:495-607), DDPM-{a.subsample}, full-size network, synthetic clips / seeded weights",
        "seconds": round(el, 2), "jobs": jobs, "jobs_per_s": round(jobs / el, 3),
        "decoded_frames_per_s": round(jobs * 30 / el, 2),
        "jobs_below_1bpp": kept, "generated_frames_kept": gen,
        "generation_rounds": stats.get("rounds"), "generation_launches": launches,
        "launch_size_histogram": {str(k): v for k, v in sorted(stats["launch_sizes"].items())},
        "score_network_forwards": fwd, "sample_forwards": samples,
        "sample_forwards_per_s": round(samples / el, 1),
        "key_frames_coded": stats.get("key_frames_coded"),
        "thresholds_psnr_db": [round(t, 2) for t in thresholds],
        "range_events": L.range_events(),
        "note": "the reference would run these jobs one at a time at B=1 per launch; here up to max_batch jobs share a launch"}))


if __name__ == "__main__":
    main()
