"""GPU test of the drop-in CLI (city_sender.py -> evc_amd/cli.py) with seeded stand-in weights and clips:
reference flags, reference output file names, the receiver loop and both policy branches."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_cli_end_to_end_synthetic(tmp_path, monkeypatch):
    import evc_amd  # noqa: F401
    from evc_amd import cli
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    monkeypatch.chdir(tmp_path)
    out = tmp_path / "out"
    base = ["--config", os.path.join(repo, "configs", "mine.yml"), "--synthetic", "--exp", str(tmp_path / "exp"),
            "--data_npy", "missing.npy", "--output_path", str(out), "--start_idx", "0", "--end_idx", "2", "--batch", "2",
            "--subsample", "2", "--q", "3", "--config_mod", "model.ngf=32 model.n_head_channels=32"]
    cli.main(base + ["--bitstream-dir", str(tmp_path / "bits")])   # fixed mask: 2 key frames + 28 generated
    d = out / "output_0"
    arr = np.load(d / "city_output_npy_idx0_q3_thr0.00.npy")  # function.py:41-52 naming: gt stacked over decoded
    assert arr.shape == (2 * 128, 30 * 128, 3) and np.isfinite(arr).all() and arr.min() >= 0 and arr.max() <= 1
    for v in (1, 2):     # three clips in batches of 2 + 1
        assert os.path.exists(out / f"output_{v}" / f"city_output_npy_idx{v}_q3_thr0.00.npy")
    bpp = np.load(d / "bpp_0.npy")
    psnr = np.load(d / "psnr_0.npy")
    assert bpp.shape == (1,) and 0 < bpp[0] and psnr.shape == (1, 30)
    assert os.path.exists(tmp_path / "exp" / "video_samples" / "arg_config" / "config.yml")
    # the receiver decoded from the container file: its payload is exactly the bits the sender reported
    from evc_amd import container
    d_rx, keys_rx, _ = container.unpack((tmp_path / "bits" / "clips_0_1_q3.evc").read_bytes())
    assert d_rx.sum() == 2 and len(d_rx) == 30
    bpp1 = np.load(out / "output_1" / "bpp_1.npy")
    assert container.payload_bits(keys_rx) == round(float(bpp[0] + bpp1[0]) * 128 * 128 * 30)
    assert os.path.exists(tmp_path / "bits" / "clips_2_2_q3.evc")
    # PSNR policy: an unreachable threshold rejects every generated frame -> everything is key-coded (more bits);
    # a trivially low one accepts everything -> same mask as above
    base[base.index("--end_idx") + 1] = "0"
    cli.main(base + ["--policy", "psnr", "--thresholds", "200", "-100"])
    bpp2 = np.load(d / "bpp_0.npy")
    assert bpp2.shape == (2,) and bpp2[0] > 5 * bpp2[1] and abs(bpp2[1] - bpp[0]) / bpp[0] < 0.2
    psnr2 = np.load(d / "psnr_0.npy")
    assert psnr2.shape == (2, 30)
