"""Command line with the reference's surface (city_sender.py:47-223, 467-617): same flags, same
``configs/mine.yml`` schema and ``--config_mod`` grammar, same checkpoint layouts, same output file names.

    python city_sender.py --data_npy data_npy/city_bonn.npy --output_path out/ --start_idx 0 --end_idx 8

Differences, all additive: ``--q`` selects ELIC quality indexes (the reference hard-codes 4 and 5,
city_sender.py:504), ``--sampler DDPM|DDIM|FPNDM``, ``--policy mask|psnr|lpips`` (+ ``--thresholds``, ``--metric``):
``lpips`` is the reference's rule (``decide_5to5_lpips``, city_sender.py:376-406, thresholds 0.30 ... 0.03 x q in {4, 5}) and
the DEFAULT whenever a perceptual metric is available -- ``--metric`` (weight files of the HIP LPIPS-AlexNet or
``pkg.module:callable``) or LPIPS weight files found where the reference keeps them (``find_lpips_weights``); without one
the explicit fallback ``mask`` (fixed transmit mask) runs and the CLI says so; ``psnr`` is the reference's own
``decide_5to5`` (city_sender.py:353-374); ``--synthetic`` builds seeded stand-ins when checkpoints / data are
absent.  ``--gpus N`` (or ``torch.distributed.run``) block-shards the video range over N ranks, one GPU each.
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

DEFAULT_PATHS = [f"checkpoints/neural network/{i}.pth.tar" for i in range(6)]


def build_parser():
    p = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    # --- the reference's flags (city_sender.py:50-130) ---
    p.add_argument("--config", type=str, default="configs/mine.yml", help="Path to the config file")
    p.add_argument("--seed", type=int, default=1234, help="Random seed")
    p.add_argument("--exp", type=str, default="checkpoints/sender", help="Path for saving running related data.")
    p.add_argument("--ni", default=True, action="store_true", help="No interaction")
    p.add_argument("--video_gen", default=True, action="store_true")
    p.add_argument("-v", "--video_folder", type=str, default="arg_config")
    p.add_argument("--subsample", type=int, default=None, help="override config.sampling.subsample")
    p.add_argument("--ckpt", type=int, default=900000, help="Model checkpoint # to load from")
    p.add_argument("--config_mod", nargs="*", type=str, default="model.ngf=192 model.n_head_channels=192")
    p.add_argument("--data_npy", type=str, default="city_bonn.npy", help="data_npy path, shape = B, T, C, H, W")
    p.add_argument("--output_path", type=str, default="test_out/", help="result output path")
    p.add_argument("-c", "--entropy-coder", choices=["ans"], default="ans", help="entropy coder")
    p.add_argument("--cuda", default=True, help="enable the GPU (always on: there is no CPU path)")
    p.add_argument("--plot", default=True, help="kept for compatibility (RD plots are out of scope)")
    p.add_argument("--entropy-estimation", action="store_true")
    p.add_argument("-p", "--path", dest="paths", type=str, nargs="+", default=DEFAULT_PATHS, help="ELIC checkpoints")
    p.add_argument("--patch", type=int, default=64, help="padding patch size")
    p.add_argument("--start_idx", type=int, default=0, help="Start video index")
    p.add_argument("--end_idx", type=int, default=0, help="End video index (inclusive, city_sender.py:495)")
    # --- additions ---
    p.add_argument("--q", type=int, nargs="+", default=[4, 5], help="ELIC quality indexes (reference loop: 4 5)")
    p.add_argument("--sampler", default="DDPM", choices=["DDPM", "DDIM", "FPNDM"])
    p.add_argument("--policy", default=None, choices=["mask", "psnr", "lpips"],
                   help="default: lpips -- the reference's rule (decide_5to5_lpips over thresholds 0.30 ... 0.03 x q in {4, 5}, "
                        "city_sender.py:376-406,504-548) -- when --metric is given or LPIPS weight files are found "
                        "(weights/v0.1/alex.pth + an alexnet-owt-*.pth backbone beside it or in the torch hub cache); "
                        "otherwise the explicit fallback `mask` (fixed transmit mask).  The rule that ran is printed.")
    p.add_argument("--thresholds", type=float, nargs="+", default=None,
                   help="psnr policy: dB thresholds; lpips policy: distances (default: the reference's sweep "
                        "0.30, 0.29 ... 0.03, city_sender.py:508)")
    p.add_argument("--metric", type=str, default=None,
                   help="lpips policy: weight files of the HIP LPIPS-AlexNet, 'alexnet-owt-*.pth,alex.pth' (torchvision backbone + "
                        "lpips v0.1 linear layers) or one saved LPIPS state dict; or package.module:callable, fn(pred, gt) -> "
                        "distances for (n,3,H,W) tensors in [0,1]")
    p.add_argument("--policy-batch", type=int, default=32,
                   help="psnr / lpips policy: (video, q, threshold) jobs stacked per score-network launch")
    p.add_argument("--bpp-limit", type=float, default=1.0,
                   help="psnr / lpips policy: a (video, q) sweep stops at the first threshold whose rate reaches this "
                        "many bits per pixel (city_sender.py:563-564)")
    p.add_argument("--gpus", type=int, default=1,
                   help="ranks (one per GPU) to shard the video range over; > 1 outside torchrun starts them itself")
    p.add_argument("--synthetic", action="store_true", help="seeded stand-ins for missing checkpoints / data")
    p.add_argument("--batch", type=int, default=8,
                   help="mask policy: clips decoded together per GPU (the reference runs one clip at a time)")
    p.add_argument("--groups", type=int, default=1, help="concurrent clip groups (HIP streams) inside a batch")
    p.add_argument("--bitstream-dir", type=str, default=None,
                   help="mask policy: write each batch's key-frame strings + mask as an EVC1 container here and "
                        "decode from the bytes read back (container.py)")
    return p


def find_lpips_weights(roots=(".",)):
    """The LPIPS weight files the reference's rule needs, where the reference keeps / fetches them: the trained linear
    layers ``weights/v0.1/alex.pth`` (shipped in the reference tree, also under models/ and benchmark/) and torchvision's
    AlexNet backbone ``alexnet-owt-*.pth`` (beside it, or in the torch hub cache where ``lpips.LPIPS(net='alex')`` downloads
    it).  -> "backbone.pth,alex.pth" (the ``--metric`` spec of the HIP LPIPS network) or None."""
    import glob as _glob
    for root in roots:
        for sub in ("weights/v0.1", "models/weights/v0.1", "benchmark/weights/v0.1"):
            lin = os.path.join(root, sub, "alex.pth")
            if not os.path.isfile(lin):
                continue
            hub = os.path.join(os.environ.get("TORCH_HOME", os.path.expanduser("~/.cache/torch")), "hub", "checkpoints")
            for d in (os.path.join(root, sub), hub):
                back = sorted(_glob.glob(os.path.join(d, "alexnet-owt-*.pth")))
                if back:
                    return f"{back[0]},{lin}"
    return None


def resolve_policy(args, log=print):
    """The decision rule of this run.  An explicit ``--policy`` wins; otherwise the reference's own rule (LPIPS thresholds
    0.30 ... 0.03, city_sender.py:376-406,504-548) whenever a perceptual metric is available, else ``mask``."""
    if args.policy is None:
        if args.metric is None:
            args.metric = find_lpips_weights()
        args.policy = "lpips" if args.metric else "mask"
        why = (f"metric {args.metric}" if args.metric else
               "no --metric and no LPIPS weight files (weights/v0.1/alex.pth + alexnet-owt-*.pth) found: explicit fallback")
        log(f"decision rule: {args.policy} ({why})")
    else:
        log(f"decision rule: {args.policy} (--policy)")
    return args.policy


class NumericsError(RuntimeError):
    pass


def check_numerics(frames, where):
    """The fp16-split arithmetic clamps nothing (include/evc_hip.h EVC_RANGE_*): an operand beyond fp16's range becomes NaN
    and the sticky range-event word says so.  Never write such frames: stop with the remedy."""
    from . import lib as L
    ev = L.range_events(reset=True)
    finite = bool(torch.isfinite(frames).all()) if torch.is_tensor(frames) else bool(np.isfinite(frames).all())
    if ev or not finite:
        raise NumericsError(
            f"{where}: range-event word {ev:#x}, frames finite: {finite}.  A GroupNorm-ed or moment-bounded operand of the "
            f"score network left fp16's range (or a tensor held NaN / inf) under the default f16x3 arithmetic; rerun with "
            f"EVC_CONV_ARITH=bf16x6 (exact 3-way bf16 split, no range assumption; about half the throughput) or =f32")


def cal_psnr(a, b, maxvalue=1.0):
    """city_sender.py:257-260."""
    mse = np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)
    return 10 * np.log10((maxvalue ** 2) / mse)


def save_output(gt, xge, q, thr, idx, output_dir):
    """function.py:41-52 (npy always; png when PIL is importable -- the reference uses cv2)."""
    os.makedirs(output_dir, exist_ok=True)
    output = np.concatenate([gt, xge], axis=0)
    np.save(os.path.join(output_dir, "city_output_npy_idx%d_q%d_thr%.2f.npy" % (idx, q, thr)), output)
    try:
        from PIL import Image
        Image.fromarray((output * 255).astype(np.uint8)).save(
            os.path.join(output_dir, "city_idx%d_q%d_thr%.2f.png" % (idx, q, thr)))
    except Exception:
        pass


def main(argv=None):
    args = build_parser().parse_args(argv)
    import yaml
    from . import dist as D
    if D.needs_self_launch(args.gpus):      # parent: starts the ranks before any HIP call, relays their exit status
        script = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "city_sender.py")
        sys.exit(D.self_launch(script, sys.argv[1:] if argv is None else list(argv), args.gpus))
    from . import ckpt, config as C, lib as L, sampler as S, synthetic
    from .decoder import ClipDecoder
    from .elic import ElicModel, inference
    from .scorenet import build_score_network

    resolve_policy(args, log=lambda m: print(m, flush=True))
    cfg, raw = C.load_config(args.config, args.config_mod)
    if args.subsample is not None:
        cfg.sampling.subsample = args.subsample
    cfg.sampling.ckpt_id = args.ckpt or cfg.sampling.ckpt_id
    rank, world, device = D.init()
    if args.gpus > 1 and world != args.gpus:
        sys.exit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    L.hip_lib()
    torch.manual_seed(args.seed)
    np.random.seed(args.seed)
    vf = os.path.join(args.exp, "video_samples", args.video_folder)
    if rank == 0:
        os.makedirs(vf, exist_ok=True)
        with open(os.path.join(vf, "config.yml"), "w") as f:
            yaml.dump(raw, f, default_flow_style=False)
        with open(os.path.join(vf, "args.yml"), "w") as f:
            yaml.dump(vars(args), f, default_flow_style=False)

    # ---- weights: rank 0 reads (or synthesises) them, one RCCL broadcast each ----
    sd_d, sd_e = None, {}
    if rank == 0:
        ck = os.path.join(args.exp, f"checkpoint_{cfg.sampling.ckpt_id}.pt")
        if os.path.exists(ck):
            sd_d = ckpt.load_diffusion_checkpoint(ck, ema=cfg.model.ema)
        elif args.synthetic:
            sd_d = synthetic.diffusion_state_dict(cfg, args.seed)
        else:
            sys.exit(f"missing {ck} (pass --synthetic for seeded stand-in weights)")
        for q in args.q:
            pth = args.paths[q]
            if os.path.exists(pth):
                sd_e[q] = ckpt.load_elic_state_dict(pth)
            elif args.synthetic:
                sd_e[q] = synthetic.elic_state_dict(q)
            else:
                sys.exit(f"missing {pth} (pass --synthetic)")
    sd_d = D.broadcast_state_dict(sd_d, 0, device, world)
    net = build_score_network(cfg, sd_d, device=device)     # model.arch: unetmore (default, + spade) | unetmorepseudo3d | unet
    models = {q: ElicModel(D.broadcast_state_dict(sd_e.get(q), 0, device, world), device=device) for q in args.q}

    if os.path.exists(args.data_npy):
        data = np.load(args.data_npy, mmap_mode="r")
    elif args.synthetic:
        data = synthetic.make_clips(args.end_idx + 1, seed=args.seed)
    else:
        sys.exit(f"missing {args.data_npy} (pass --synthetic)")

    lo, hi = D.shard_range(args.end_idx + 1 - args.start_idx, rank, world)
    gen = torch.Generator(device=device).manual_seed(args.seed + rank)
    if args.policy == "lpips":       # city_sender.py:508: np.arange(0.30, 0.02, -0.01) rounded to 2 decimals
        thresholds = args.thresholds or [float("%.2f" % t) for t in np.arange(0.30, 0.02, -0.01)]
    else:
        thresholds = args.thresholds if args.policy == "psnr" and args.thresholds else [0.0]
    t_start = time.time()
    vids = list(range(args.start_idx + lo, args.start_idx + hi))

    def report(vid, q, thr, x, gt, bits, d, store):
        bpp = sum(bits) / 128 / 128 / 30
        ps = [cal_psnr(x[i], gt[i]) for i in range(30)]
        print(f"[rank {rank}] video {vid} q{q} thr {thr:.2f}: d={[int(v) for v in d[:30]]} BPP {bpp:.5f} PSNR {np.mean(ps):.3f}",
              flush=True)
        store.setdefault(vid, ([], []))
        store[vid][0].append(ps); store[vid][1].append(bpp)
        g = np.concatenate(list(gt.transpose(0, 2, 3, 1)), axis=1)
        xg = np.concatenate(list(x.transpose(0, 2, 3, 1)), axis=1)
        save_output(g, xg, q, thr, vid, os.path.join(args.output_path, f"output_{vid}"))

    store = {}
    if args.policy == "mask":
        # every clip has the same transmit mask (2 key frames, then generated): decode `--batch` clips per launch
        # through the receiver (the path bench.py measures) instead of one clip at a time
        from .decoder import all_generated_mask
        from .elic import count_bits
        mask = all_generated_mask()
        for q in args.q:
            model = models[q]
            dec = ClipDecoder(net, model, cfg, S.get_sampler(args.sampler), groups=args.groups)
            for b0 in range(0, len(vids), max(1, args.batch)):
                chunk = vids[b0:b0 + max(1, args.batch)]
                gt = torch.from_numpy(np.stack([np.asarray(data[v], dtype=np.float32) / 255.0 for v in chunk]))
                keys, shape = [], None
                for f in (0, 1):                                   # key frames (city_sender.py:521-524), batched
                    pad = (-gt.shape[-1]) % args.patch, (-gt.shape[-2]) % args.patch
                    xk = torch.nn.functional.pad(gt[:, f], (0, pad[0], 0, pad[1]))
                    enc = model.compress(xk.to(device))
                    keys.append(enc["strings"]); shape = enc["shape"]
                d_rx, keys_rx, shape_rx = mask, keys, shape
                if args.bitstream_dir:                             # sender -> file -> receiver
                    from . import container
                    os.makedirs(args.bitstream_dir, exist_ok=True)
                    path = os.path.join(args.bitstream_dir, f"clips_{chunk[0]}_{chunk[-1]}_q{q}.evc")
                    with open(path, "wb") as fh:
                        fh.write(container.pack(mask, keys, shape, codec=model.codec_tag()))
                    with open(path, "rb") as fh:          # refuses a stream coded under another arithmetic
                        d_rx, keys_rx, shape_rx = container.unpack(fh.read(), expect_codec=model.codec_tag())
                frames = dec.decode(d_rx, keys_rx, shape_rx, generator=gen)[..., :gt.shape[-2], :gt.shape[-1]]
                check_numerics(frames, f"videos {chunk[0]}..{chunk[-1]} q{q}")
                x_all = frames.cpu().numpy()
                for j, vid in enumerate(chunk):
                    bits = [count_bits([[[[p[j]] for p in sl] for sl in k[0]], [k[1][j]]]) for k in keys]
                    report(vid, q, 0.0, x_all[j], gt[j].numpy(), bits, mask, store)
    else:
        # city_sender.py:495-607 batched (policy.py): every (video, q, threshold) job of this rank advances in lockstep,
        # `--policy-batch` jobs per score-network launch; key frames coded once per (video, q, frame).
        from . import policy as P
        metric = P.load_metric(args.policy, args.metric, device)
        dec = ClipDecoder(net, None, cfg, S.get_sampler(args.sampler))
        clips = {vid: torch.from_numpy(np.asarray(data[vid], dtype=np.float32) / 255.0) for vid in vids}
        res = P.run_policy(dec, models, clips, args.q, thresholds, metric, patch=args.patch, max_batch=args.policy_batch,
                           seed=args.seed, device=device, bpp_limit=args.bpp_limit, log=lambda m: print(f"[rank {rank}] {m}", flush=True))
        metric_vals = {}
        for vid in vids:
            for q in args.q:
                for r in res[(vid, q)]:
                    check_numerics(r["x"], f"video {vid} q{q} thr {r['thr']:.2f}")
                    report(vid, q, r["thr"], r["x"], clips[vid].numpy(), r["bits"], r["d"], store)
                    if args.policy == "lpips":      # per-frame distances of the decoded clip (city_sender.py:570-571)
                        v = metric.values(torch.from_numpy(r["x"]).to(device), clips[vid].to(device))
                        metric_vals.setdefault(vid, []).append(v)
        for vid, vals in metric_vals.items():
            out_root = os.path.join(args.output_path, f"output_{vid}")
            np.save(os.path.join(out_root, f"lpips_frames_{vid}.npy"), np.asarray(vals))
            np.save(os.path.join(out_root, f"lpips_{vid}.npy"),
                    P.rd_envelope(store[vid][1], np.mean(np.asarray(vals), 1), higher_is_better=False))
    for vid, (ps, bpps) in store.items():
        out_root = os.path.join(args.output_path, f"output_{vid}")
        os.makedirs(out_root, exist_ok=True)
        # reference names (function.py:148-230): psnr_<idx>.npy = RD envelope [bpp; mean PSNR] of the video's sweep;
        # the raw sweep is kept beside it (per-threshold per-frame PSNR, per-threshold bpp)
        from .policy import rd_envelope
        np.save(os.path.join(out_root, f"psnr_{vid}.npy"), rd_envelope(bpps, np.mean(np.asarray(ps), 1), True))
        np.save(os.path.join(out_root, f"psnr_frames_{vid}.npy"), np.asarray(ps))
        np.save(os.path.join(out_root, f"bpp_{vid}.npy"), np.asarray(bpps))
    D.barrier()
    if rank == 0:
        print(f"done in {time.time() - t_start:.1f}s")


if __name__ == "__main__":
    main()
