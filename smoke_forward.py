"""__graft_entry__.smoke(): one tiny full forward on cuda:0 checked against the oracle."""
import torch


def run():
    from oracle import model_ref as M
    from thinktwice_amd import model as tm, params, synth
    hw = (128, 256)
    m, cfg = tm.build_thinktwice(final_dim=hw)
    sd = params.init_params(cfg, seed=0)
    m.load_state_dict(sd)
    batch = synth.make_batch(1, img_hw=hw, num_points=8192)
    out = m.forward_inference(tm.batch_to_device(batch))
    torch.cuda.synchronize()
    with torch.no_grad():
        ref = M.forward_inference(sd, cfg, batch)
    worst = 0.0
    for k in ("pred_wp", "mu_branches", "sigma_branches", "refine_BEV_feature"):
        e = float((out[k].cpu() - ref[k]).abs().max() / ref[k].abs().max())
        worst = max(worst, e)
    assert worst < 1e-3, f"forward mismatch vs oracle: {worst}"
    print(f"smoke: full forward_inference ok (worst rel err {worst:.2e}, pred_wp[0,-1]={out['pred_wp'][0, -1].tolist()})")
