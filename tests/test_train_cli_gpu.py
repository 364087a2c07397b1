"""train.py end to end on a tiny folder dataset: data loader -> iterations -> log / test lines -> checkpoint -> resume."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_train_cli_runs_logs_saves_and_resumes(tmp_path):
    from PIL import Image
    rng = np.random.RandomState(1)
    data_dir = tmp_path / "imgs"
    data_dir.mkdir()
    for i in range(12):
        Image.fromarray(rng.randint(0, 256, size=(80, 72, 3), dtype=np.uint8)).save(data_dir / f"{i:03d}.png")
    base = [sys.executable, os.path.join(ROOT, "train.py"), "--dataset_path", str(data_dir), "--dataset_type", "normal",
            "--image_size", "64", "--batch_size", "4", "--no_dco", "--num_workers", "0", "--log_every", "2",
            "--show_every", "3", "--d_reg_every", "2"]
    env = dict(os.environ, PYTHONPATH=ROOT)
    r = subprocess.run(base + ["--exp_name", "t0", "--num_iters", "4", "--save_every", "4"], cwd=tmp_path, env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    log = (tmp_path / "experiments/t0/training_logs.txt").read_text()
    assert "[0000002/0000004] Total:" in log and "[0000004/0000004] Total:" in log
    assert "[Testing 0000003/0000004] sigma=1 delta=50%" in log and "ACC of Msg:" in log
    ckpt = tmp_path / "experiments/t0/checkpoints/0000004.pt"
    assert ckpt.exists()
    r = subprocess.run(base + ["--exp_name", "t1", "--num_iters", "6", "--save_every", "100", "--ckpt", str(ckpt)],
                       cwd=tmp_path, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    log = (tmp_path / "experiments/t1/training_logs.txt").read_text()
    assert "[0000006/0000006] Total:" in log and "[0000002/" not in log      # resumed at iteration 5
    assert (tmp_path / "experiments/t1/checkpoints/0000006.pt").exists()
