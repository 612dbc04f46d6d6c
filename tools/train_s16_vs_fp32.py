"""gradients of one training step on the S16 kernels against the exact-fp32 kernels (two processes of this script):
   python tools/train_s16_vs_fp32.py dump out.pt [batch] [size]   (honours AMMC_TRAIN_PRECISION)
   python tools/train_s16_vs_fp32.py oracle out.pt [batch] [size]     (CPU oracle in fp64: the yardstick)
   python tools/train_s16_vs_fp32.py cmp a.pt b.pt"""
import sys
sys.path.insert(0, '.')
import torch
if sys.argv[1] == "dump":
    import ammcnet_aaai2021_amd as A
    from ammcnet_aaai2021_amd import synthetic as S, harness as Hn
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    hw = int(sys.argv[4]) if len(sys.argv) > 4 else 256
    net = A.get_twostream((12, 6), (3, 2), 64, 256, 2)
    net.load_state_dict(S.make_twostream_state())
    net = net.cuda().train()
    rgb_x, op_x, rgb_t, op_t = (t.cuda() for t in S.make_clips(B, hw, hw, tag="s16-vs-fp32"))
    out = net(rgb_x, op_x)
    loss = Hn.generator_loss(out, rgb_t, op_t)
    loss.backward()
    torch.save({"loss": float(loss), "grads": {k: p.grad.cpu() for k, p in net.named_parameters()},
                "bufs": {k: v.cpu() for k, v in net.state_dict().items()}}, sys.argv[2])
elif sys.argv[1] == "oracle":
    from ammcnet_aaai2021_amd import synthetic as S
    from oracle import ammc_oracle as O
    torch.set_num_threads(32)
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    hw = int(sys.argv[4]) if len(sys.argv) > 4 else 256
    sd = {k: (v.double() if v.is_floating_point() else v) for k, v in O.clone_state(S.make_twostream_state()).items()}
    for k, v in sd.items():
        if v.is_floating_point() and k.rsplit(".", 1)[-1] not in ("running_mean", "running_var", "embed", "cluster_size", "embed_avg"):
            v.requires_grad_(True)
    rgb_x, op_x, rgb_t, op_t = (t.double() for t in S.make_clips(B, hw, hw, tag="s16-vs-fp32"))
    out = O.twostream_forward(sd, rgb_x, op_x, 2, training=True)
    loss = O.generator_loss(out, rgb_t, op_t)
    loss.backward()
    torch.save({"loss": float(loss), "grads": {k: v.grad.float() for k, v in sd.items() if v.requires_grad and v.grad is not None},
                "bufs": {}}, sys.argv[2])
else:
    a, b = torch.load(sys.argv[2]), torch.load(sys.argv[3])
    errs = []
    for k, g in a["grads"].items():
        if k not in b["grads"]:
            continue
        r = b["grads"][k].double()
        if float(r.abs().max()) > 0:
            errs.append((float((g.double() - r).norm() / r.norm()), k))
    errs.sort()
    print("loss", a["loss"], b["loss"])
    print("grad L2-rel: min %.2e median %.2e max %.2e (%s)" % (errs[0][0], errs[len(errs) // 2][0], errs[-1][0], errs[-1][1]))
    if b["bufs"]:
      be = max(float((v.double() - b["bufs"][k].double()).abs().max() / b["bufs"][k].double().abs().max().clamp_min(1e-30))
               for k, v in a["bufs"].items() if v.is_floating_point())
      print("buffers / parameters after the step: max rel diff %.2e" % be)
