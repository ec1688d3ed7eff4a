"""Closed-loop post-processing: this package's process_action / control_pid vs a 16-tick golden
sequence recorded from the reference model's own methods (stateful PID windows included)."""
import os

import numpy as np
import torch

from thinktwice_amd import config, control


def test_control_matches_reference_sequence(golden_dir):
    f = np.load(os.path.join(golden_dir, "f9_control.npz"))
    c = config.model_config()["cfg"]
    turn = control.PIDController(c["turn_KP"], c["turn_KI"], c["turn_KD"], c["turn_n"])
    spd = control.PIDController(c["speed_KP"], c["speed_KI"], c["speed_KD"], c["speed_n"])
    for t in range(f["mu"].shape[0]):
        pred = {"mu_branches": torch.from_numpy(f["mu"][t]).float(), "sigma_branches": torch.from_numpy(f["sigma"][t]).float()}
        speed = torch.from_numpy(f["speed"][t]).float()
        s1, th1, b1, meta = control.process_action(pred, 3, speed, f["target"][t])
        np.testing.assert_allclose([s1, th1, b1], f["pa"][t], rtol=0, atol=1e-12)
        s2, th2, b2, m2 = control.control_pid(c, turn, spd, torch.from_numpy(f["wp"][t]).float(), speed,
                                              f["target"][t].astype(np.float32).copy())
        got = [s2, th2, float(b2), m2["desired_speed"], m2["angle"], m2["angle_last"], m2["angle_target"],
               m2["angle_final"], m2["delta"]]
        np.testing.assert_allclose(got, f["pid"][t], rtol=0, atol=1e-12)
    assert set(meta) == {"speed", "steer", "throttle", "brake", "command", "target_point"}
