"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every
symbol include/ammc_hip.h declares; the host modules have the reference's state_dict
schema and fail loudly (no fallback) off-GPU or in modes whose kernels do not exist."""
import json
import os
import re

import ctypes as C

import pytest
import torch

import ammcnet_aaai2021_amd as A
from ammcnet_aaai2021_amd import _lib, synthetic as S
from conftest import GOLDEN, ROOT


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "ammc_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ammc_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    declared = _declared_symbols()
    assert len(declared) >= 15
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/ammc_hip.h but not exported"
        assert name in _lib.SIGNATURES, f"{name} has no ctypes signature in _lib.py"
    assert sorted(_lib.SIGNATURES) == declared
    assert lib.ammc_abi_version() == _lib.ABI_VERSION
    assert b"gfx950" in lib.ammc_build_info()
    assert lib.ammc_error_string(-1).startswith(b"invalid")


def test_argument_errors_are_status_codes_not_aborts():
    lib = _lib.load()
    assert lib.ammc_conv_gemm_f32(None, None) == -1
    d = _lib.AmmcConvDesc()
    assert lib.ammc_conv_gemm_f32(d, None) == -1                      # null pointers
    assert lib.ammc_memory_topk_blocks(0) == 0 and lib.ammc_memory_topk_blocks(129) == 5      # 32 rows per workgroup
    assert lib.ammc_maxpool2x2_f32(None, 0, 0, 0, None, 0, 0, 0, 1, 1, 1, 4, None) == -1
    with pytest.raises(_lib.AmmcHipError):
        _lib.check(-2, "x")


def test_round5_entry_points_reject_bad_arguments_without_a_gpu():
    """the entry points added in round 5 (include/ammc_hip.h): argument errors are decided before anything touches the
    device, so they are checkable here"""
    lib = _lib.load()
    # memory_topk_f16r: workspace size, block count, null / misaligned / unsupported arguments
    assert lib.ammc_codebook_f16_tiles_bytes(512, 8192) == 256 * 33 * 1024      # 256 tiles of (32 k-steps + 1 constants) KB
    assert lib.ammc_codebook_f16_tiles_bytes(500, 8192) == 0 and lib.ammc_codebook_f16_tiles_bytes(512, 0) == 0
    assert lib.ammc_memory_topk_f16r_blocks(0) == 0 and lib.ammc_memory_topk_f16r_blocks(262144) == 8192
    assert lib.ammc_pack_codebook_f16_tiles(None, 512, 8192, None, None) == -1
    assert lib.ammc_pack_codebook_f16_tiles(64, 500, 8192, 64, None) == -1       # d % 16
    assert lib.ammc_pack_codebook_f16_tiles(64, 512, 8192, 72, None) == -1       # tiles not 16-byte aligned
    args = [64, 64, 64, 1024, 512, 8192, 2, 64, 64, 64, 64, None]
    assert lib.ammc_memory_topk_fwd_f16r(*([None] + args[1:])) == -1
    for pos, val, want in ((3, 0, -1), (6, 0, -1), (6, 9000, -1), (4, 64, -2), (4, 1024, -2), (6, 5, -2), (0, 72, -1)):
        bad = list(args)
        bad[pos] = val
        assert lib.ammc_memory_topk_fwd_f16r(*bad) == want, (pos, val)
    # conv_first_s16_bs: batch-strided NCHW input
    first = [64, 12 * 256 * 256, 16, 12, 256, 256, 64, None, None, 1, 64, 8, 8, 8, None, None]
    assert lib.ammc_conv_first_s16_bs(*([None] + first[1:])) == -1
    for pos, val, want in ((1, -1, -1), (3, 17, -2), (5, 250, -2), (9, 2, -2), (10, 72, -1), (11, 4, -1)):
        bad = list(first)
        bad[pos] = val
        assert lib.ammc_conv_first_s16_bs(*bad) == want, (pos, val)
    # weight-gradient slabs: no slab form -> 0 floats; the launch entry refuses null / short workspaces
    d = _lib.AmmcWgradDesc()
    assert lib.ammc_conv_wgrad_s16_slab_floats(None) == 0 and lib.ammc_conv_wgrad_s16_slab_floats(C.byref(d)) == 0
    assert lib.ammc_conv_wgrad_s16_slabs(C.byref(d), None, None, 0, None, 64, 64, None) == -1
    assert lib.ammc_scale_shift_act_s16_pool_supported(60, 256, 256, 258 * 64, 64, 258 * 64, 64, 64) == 0    # c % 8
    assert lib.ammc_scale_shift_act_s16_pool_supported(64, 255, 256, 258 * 64, 64, 258 * 64, 64, 64) == 0   # odd height
    assert lib.ammc_scale_shift_act_s16_pool_supported(64, 256, 256, 258 * 64, 64, 258 * 64, 64, 64) == 1
    digests = lib.ammc_source_digests().decode()
    assert "memory_topk_f16r.hip" in digests and "ammc_common.h" in digests


def test_round6_entry_points_reject_bad_arguments_without_a_gpu():
    """`ammc_memory_block_s16`, `ammc_pack_frag_rows_s16` and the guarded packs: argument errors are status codes decided
    before anything touches the device"""
    lib = _lib.load()
    ok = dict(x=64, xs=(34 * 34 * 512, 34 * 512, 512), y=128, ys=(34 * 34 * 512, 34 * 512, 512), b=2, h=32, w=32, c=512,
              enc_w=256, enc_b=512, e16=1024, emd=2048, en=4096, d=64, m=2000, k=2, dec_w=8192, dec_b=16384, idx=64, qk=None,
              q1=None, part=64, diff=64, cnt=64, flag=None)

    def call(**kw):
        a = dict(ok, **kw)
        return lib.ammc_memory_block_s16(a["x"], *a["xs"], a["y"], *a["ys"], a["b"], a["h"], a["w"], a["c"], a["enc_w"], a["enc_b"],
                                         a["e16"], a["emd"], a["en"], a["d"], a["m"], a["k"], a["dec_w"], a["dec_b"], a["idx"],
                                         a["qk"], a["q1"], a["part"], a["diff"], a["cnt"], a["flag"], None)
    assert call(x=None) == -1 and call(cnt=None) == -1 and call(b=0) == -1
    assert call(xs=(34 * 34 * 512, 34 * 512, 510)) == -1          # strides: whole S16 groups
    assert call(x=72) == -1                                       # activations 32-byte aligned
    for kw in (dict(c=256), dict(d=128), dict(k=1), dict(m=4096)):
        assert call(**kw) == -2, kw                               # not the shipped block's shape: the five-launch chain
    assert lib.ammc_pack_frag_rows_s16(None, 512, 128, 64, None) == -1
    assert lib.ammc_pack_frag_rows_s16(64, 500, 128, 128, None) == -1 and lib.ammc_pack_frag_rows_s16(64, 512, 100, 128, None) == -1
    assert lib.ammc_split_rows_guarded_f32(None, 64, 64, None, None) == -1 and lib.ammc_split_rows_guarded_f32(64, 60, 128, None, None) == -1
    assert lib.ammc_pack_codebook_s16_guarded(None, 64, 256, 64, None, None) == -1
    assert lib.ammc_pack_codebook_s16_guarded(64, 60, 256, 64, None, None) == -1


def test_state_dict_schema_matches_reference():
    with open(os.path.join(GOLDEN, "param_counts.json")) as fp:
        pc = json.load(fp)
    net = A.get_twostream((12, 6), (3, 2), 64, 256, 2)
    assert list(net.state_dict().keys()) == pc["twostream_state_keys"]
    assert sum(p.numel() for p in net.parameters()) == pc["twostream"] == 25049029
    assert sum(p.numel() for p in A.get_unet_vq_topk_res(12, 3, 64, 256, 2).parameters()) == 7805891
    assert sum(p.numel() for p in A.get_unet(12, 3).parameters()) == pc["unet_12_3"]
    sd = S.make_twostream_state()
    res = net.load_state_dict(sd, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    q = net.rgb.vq_down3.quan.quantize
    assert (q.dim, q.n_embed, q.k, q.decay, q.eps) == (64, 256, 2, 0.99, 1e-5)


def test_no_silent_fallback():
    net = A.get_twostream((12, 6), (3, 2), 64, 256, 2).eval()
    with pytest.raises(_lib.AmmcHipError):
        net(torch.zeros(1, 12, 64, 64), torch.zeros(1, 6, 64, 64))          # CPU tensors
    with pytest.raises(_lib.AmmcHipError):
        net.rgb.inc(torch.zeros(1, 12, 16, 16))
    net.train()
    with pytest.raises(_lib.AmmcHipError):                                   # training mode: HIP only as well
        net(torch.zeros(1, 12, 64, 64), torch.zeros(1, 6, 64, 64))
    with pytest.raises(_lib.AmmcHipError):                                   # stand-alone conv block in .train(): HIP only too
        net.rgb.inc(torch.zeros(1, 12, 16, 16))
    with pytest.raises(_lib.AmmcHipError):                                   # the memory block alone in .train(): HIP only too
        net.rgb.vq_down3(torch.zeros(1, 512, 4, 4))


def test_synthetic_data_is_deterministic():
    a = S.hashed_uniform("x", (5, 7))
    b = S.hashed_uniform("x", (5, 7))
    assert torch.equal(a, b) and float(a.abs().max()) <= 1.0
    # pinned values: the golden fixtures depend on this generator never changing
    assert abs(float(S.hashed_uniform("pin", (3,))[1]) - float(S.hashed_uniform("pin", (3,))[1])) == 0.0
    rgb, op, rt, ot = S.make_clips(2, 16, 16)
    assert rgb.shape == (2, 12, 16, 16) and op.shape == (2, 6, 16, 16) and ot.shape == (2, 2, 16, 16)
    assert torch.allclose(op[:, 1::2], op[:, 0::2] / 256.0)


def _fake_desc(B, H, W, cin, n, ntaps=9, up=1, y_f32=0):
    """descriptor with fake (aligned, never dereferenced) addresses: `ammc_conv_gemm_s16_variant` runs the argument
    checks and the dispatch of `ammc_conv_gemm_s16` and launches nothing, so it needs no GPU"""
    d = _lib.AmmcConvDesc()
    d.x, d.w, d.y = 0x100000, 0x200000, 0x300000
    d.batch, d.height, d.width, d.cin, d.ntaps, d.n, d.up = B, H, W, cin, ntaps, n, up
    d.cgroup = n // 4 if up == 2 else n
    d.y_f32 = y_f32
    d.x_ps, d.x_rs = cin, (W + 2) * cin
    d.x_bs = (H + 2) * d.x_rs
    d.y_ps, d.y_rs = n, (W * up + 2) * n
    d.y_bs = (H * up + 2) * d.y_rs
    return d


def test_statistics_rows_of_the_training_forward_layers():
    """which 3x3 layers of the timed training step (batch 32, 256x256: BASELINE.json configs[2]) get their BatchNorm
    statistics from the convolution's own epilogue (`AmmcConvDesc.stats`): the k-half-major halo-patch kernels of the
    256x256, 128x128 and 64x64 levels; the 16x16x32 kernel of the 32x32 level and the implicit-GEMM kernel of the first
    layer report 0 rows"""
    from ammcnet_aaai2021_amd.engine import s16_variant
    lib = _lib.load()
    B = 32
    for (hw, cin, n), rows in {(256, 64, 64): B * 32 * 8, (256, 128, 64): B * 32 * 8, (128, 64, 128): B * 16 * 4,
                               (128, 256, 128): B * 16 * 4, (64, 128, 256): B * 8 * 2, (32, 512, 512): 0, (256, 16, 64): 0}.items():
        d = _fake_desc(B, hw, hw, cin, n, y_f32=1)
        assert lib.ammc_conv_gemm_s16_stats_rows(C.byref(d)) == rows, (hw, cin, n)
        d.stats = 0x500000
        if rows:
            assert s16_variant(d).endswith(", 0, 1>+stats")
        else:
            assert lib.ammc_conv_gemm_s16_variant(C.byref(d), C.create_string_buffer(96), 96) == -2      # AMMC_EUNSUP
    d = _fake_desc(B, 256, 256, 64, 64, y_f32=0)                 # an S16 output has no statistics epilogue
    assert lib.ammc_conv_gemm_s16_stats_rows(C.byref(d)) == 0


def test_s16_dispatch_of_the_benchmark_shapes():
    """which kernel each layer of the benchmark's workload (batch 16, 256x256: BASELINE.json configs[1]) gets - the
    variants named here are the ones tests/test_gpu_conv_tap.py and the batch-16 golden test must reach"""
    from ammcnet_aaai2021_amd.engine import s16_variant
    B = 16
    want = {
        (256, 64, 64): "conv_tap_s16<4, 1, 2, 2, 1, 0, 1>",       # inc.1, up3.1: the k-half-major pipeline (KH), 64 filters
        (256, 128, 64): "conv_tap_s16<4, 1, 2, 2, 1, 0, 1>",      # up3.0
        (128, 64, 128): "conv_tap_s16<4, 1, 2, 4, 1, 0, 1>",      # down1.0 ... KH, 128 filters (two rounds of 512 workgroups and up)
        (128, 128, 128): "conv_tap_s16<4, 1, 2, 4, 1, 0, 1>",
        (128, 256, 128): "conv_tap_s16<4, 1, 2, 4, 1, 0, 1>",
        (64, 128, 256): "conv_tap_s16<4, 2, 2, 2, 2, 1>",         # 512 tiles = one round of two workgroups per CU: the 8-wave variant
        (64, 256, 256): "conv_tap_s16<4, 2, 2, 2, 2, 1>",
        (64, 512, 256): "conv_tap_s16<4, 2, 2, 2, 2, 1>",
        (32, 256, 512): "conv_tap_s16<4, 2, 2, 2, 2, 1>",         # 256 tiles: the 8-wave variant
        (32, 512, 512): "conv_tap_s16<4, 2, 2, 2, 2, 1>",
    }
    for (hw, cin, n), label in want.items():
        assert s16_variant(_fake_desc(B, hw, hw, cin, n)) == label, (hw, cin, n)
    outc = _fake_desc(B, 256, 256, 64, 32, y_f32=1)                                                    # outc: 3 real filters
    outc.n_store, outc.act, outc.y_ps, outc.y_rs, outc.y_cs, outc.y_bs = 3, 2, 1, 256, 256 * 256, 3 * 256 * 256
    assert s16_variant(outc) == "conv_outc_s16"
    assert s16_variant(_fake_desc(B, 256, 256, 16, 64)).startswith("conv_gemm_s16<")                   # inc.0 (12 -> 16 channels)
    assert s16_variant(_fake_desc(B, 32, 32, 512, 1024, ntaps=1, up=2)).startswith("conv_gemm_s16<")   # ConvTranspose
    # small batches fall back to the implicit GEMM (split-K below 192 tiles)
    d = _fake_desc(1, 32, 32, 512, 512)
    assert s16_variant(d) == "conv_gemm_s16<128x128>"
    d.splitk_ws, d.splitk_ws_floats = 0x400000, 8 << 20
    assert s16_variant(d).startswith("conv_gemm_s16<128x128>+splitk")
    # per-CALL choices through the descriptor (re-entrant: no process state is touched)
    dd = _fake_desc(B, 128, 128, 128, 128)
    dd.s16_mf = 2
    assert s16_variant(dd) == "conv_tap_s16<4, 1, 2, 4, 1, 1>"
    dd.s16_mf = 1
    assert s16_variant(dd) == "conv_tap_s16<4, 1, 2, 4, 1, 0>"
    assert s16_variant(_fake_desc(B, 128, 128, 128, 128)) == "conv_tap_s16<4, 1, 2, 4, 1, 0, 1>"      # the default is untouched
    outc.outc_stream = 1
    assert s16_variant(outc) == "conv_tap_s16<4, 1, 2, 1, 1, 1>"
    outc.outc_stream = 0
    assert s16_variant(outc) == "conv_outc_s16"
    dd.s16_mf = 3
    assert _lib.load().ammc_conv_gemm_s16_variant(C.byref(dd), C.create_string_buffer(64), 64) == -1     # AMMC_EINVAL
    lib = _lib.load()                                                        # the process-wide A/B switch of the MFMA shape
    assert lib.ammc_set_option(b"s16_mf", 1) == 0
    assert s16_variant(_fake_desc(B, 128, 128, 128, 128)) == "conv_tap_s16<4, 1, 2, 4, 1, 1>"
    assert lib.ammc_set_option(b"s16_mf", 0) == 0
    assert s16_variant(_fake_desc(B, 256, 256, 64, 64)) == "conv_tap_s16<4, 1, 2, 2, 1, 0>"
    assert lib.ammc_set_option(b"s16_mf", -1) == 0 and lib.ammc_set_option(b"s16_mf", 2) == -1
    assert lib.ammc_set_option(b"outc_stream", 0) == 0 and lib.ammc_set_option(b"outc_stream", 2) == -1
    assert s16_variant(outc) == "conv_tap_s16<4, 1, 2, 1, 1, 1>"             # the halo-patch kernel's output-layer instance
    assert lib.ammc_set_option(b"outc_stream", 1) == 0
    assert lib.ammc_set_option(b"no_such_option", 1) == -2
    bad = _fake_desc(1, 32, 32, 24, 64)                                      # cin not a power of two
    with pytest.raises(_lib.AmmcHipError):
        s16_variant(bad)


def test_bench_caps_the_rccl_channels_for_training_ranks_only_on_request():
    """bench.py --mode train --gpus N --rccl-channels 4: NCCL_MAX_NCHANNELS is set before the process group exists (DESIGN.md
    section 6 models 4 as the better setting; no multi-GPU A/B has measured it, so RCCL's own default stays the default -
    round-5 advisor), an explicit environment setting wins, inference / stress ranks and the gloo test mode are left alone"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    env = {}
    assert b.default_rccl_channels("train", False, env) is None and env == {}              # RCCL's default unless asked
    assert b.default_rccl_channels("train", False, env, channels=4) == "4" and env == {"NCCL_MAX_NCHANNELS": "4"}
    env = {"NCCL_MAX_NCHANNELS": "16"}
    assert b.default_rccl_channels("train", False, env, channels=4) == "16"
    for mode, share in (("infer", False), ("stress", False), ("train", True)):
        env = {}
        assert b.default_rccl_channels(mode, share, env, channels=4) is None and env == {}
    topo = b.host_topology()
    assert topo and all(len(v) >= 1 for v in topo.values())


def test_header_compiles_as_c_and_agrees_with_the_ctypes_mirror(tmp_path):
    """the boundary is a C ABI: include/ammc_hip.h must compile as plain C99 (what a cgo / JNI / FFI binding of the
    reference's side would include), a C program must link against the library and get status codes back, and the two
    descriptor structs must have the size and field offsets `_lib.py` mirrors with ctypes (a silent disagreement would
    scramble every launch)"""
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no C compiler")
    lib = _lib.load()
    fields = {"AmmcConvDesc": [f[0] for f in _lib.AmmcConvDesc._fields_],
              "AmmcWgradDesc": [f[0] for f in _lib.AmmcWgradDesc._fields_]}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "ammc_hip.h"', 'int main(void) {',
             '  printf("abi %d\\n", ammc_abi_version());',
             '  printf("einval %d\\n", ammc_conv_gemm_f32(NULL, NULL));',
             '  printf("blocks %d\\n", ammc_memory_topk_blocks(129));',
             '  printf("err %s\\n", ammc_error_string(-2));']
    for name, fl in fields.items():
        lines.append(f'  printf("sizeof {name} %zu\\n", sizeof({name}));')
        for f in fl:
            lines.append(f'  printf("off {name}.{f} %zu\\n", offsetof({name}, {f}));')
    lines += ['  return 0;', '}']
    src = tmp_path / "cabi.c"
    src.write_text("\n".join(lines) + "\n")
    libdir = os.path.dirname(_lib.LIB_PATH) if hasattr(_lib, "LIB_PATH") else os.path.join(ROOT, "ammcnet_aaai2021_amd")
    exe = tmp_path / "cabi"
    cmd = [gcc, "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), str(src),
           "-o", str(exe), "-L", libdir, "-lammc_hip", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]
    got = dict(ln.rsplit(" ", 1) for ln in out.stdout.strip().splitlines() if not ln.startswith("err "))
    assert int(got["abi"]) == _lib.ABI_VERSION == lib.ammc_abi_version()
    assert int(got["einval"]) == -1 and int(got["blocks"]) == 5
    assert "err unsupported" in out.stdout or "err " in out.stdout
    for name, cls in (("AmmcConvDesc", _lib.AmmcConvDesc), ("AmmcWgradDesc", _lib.AmmcWgradDesc)):
        assert int(got[f"sizeof {name}"]) == C.sizeof(cls), name
        for f in fields[name]:
            assert int(got[f"off {name}.{f}"]) == getattr(cls, f).offset, (name, f)
