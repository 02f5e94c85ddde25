"""CPU-only checks of the drop-in boundary: the C-ABI library loads and exports every symbol include/*.h declares,
the ctypes structs match the header layout, and the host mirror keeps the reference's state_dict contract."""
import ctypes
import os
import re

import pytest
import torch

from selfmask_amd import _native as N
from selfmask_amd import MaskFormer, get_model, BaseStructure, state_shapes, synthetic_state_dict

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(REPO, "include", "selfmask_hip.h")).read()
    declared = set(re.findall(r"\b(sm_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no prototypes found in the header"
    lib = N.load()
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in selfmask_hip.h but not exported by the .so"
    assert declared == set(N.SYMBOLS), (declared ^ set(N.SYMBOLS))
    assert lib.sm_version() >= 100


def test_struct_sizes_match_header_layout(tmp_path):
    """sizeof()/offsetof() as the C compiler sees the header vs the ctypes mirrors (gcc is part of the image)."""
    import shutil
    import subprocess
    cc = shutil.which("gcc") or shutil.which("cc")
    if cc is None:
        pytest.skip("no C compiler")
    pairs = {"sm_gemm_args": N.GemmArgs, "sm_ln_args": N.LnArgs, "sm_attn_args": N.AttnArgs, "sm_weights": N.Weights,
             "sm_forward_io": N.ForwardIO, "sm_eval_args": N.EvalArgs, "sm_eval_image": N.EvalImage,
             "sm_bilateral_args": N.BilateralArgs, "sm_enc_layer": N.EncLayer, "sm_dec_layer": N.DecLayer,
             "sm_row_map": N.RowMap, "sm_kernel_time": N.KernelTime, "sm_qkv_attn_args": N.QkvAttnArgs, "sm_pre_image": N.PreImage,
             "sm_spectral_args": N.SpectralArgs}
    last = {"sm_gemm_args": "mfma_terms", "sm_ln_args": "chain_eps", "sm_attn_args": "scale", "sm_weights": "no_objectness",
            "sm_forward_io": "last_layer_only", "sm_eval_args": "scale", "sm_bilateral_args": "W", "sm_qkv_attn_args": "mfma_terms", "sm_pre_image": "ksy",
            "sm_spectral_args": "kmeans_max_iter"}
    src = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{os.path.join(REPO, "include", "selfmask_hip.h")}"',
           'int main(void){']
    for cname in pairs:
        src.append(f'printf("{cname} %zu\\n", sizeof({cname}));')
    for cname, field in last.items():
        src.append(f'printf("{cname}.{field} %zu\\n", offsetof({cname}, {field}));')
    src.append('return 0;}')
    c = tmp_path / "sz.c"
    c.write_text("\n".join(src))
    exe = tmp_path / "sz"
    subprocess.run([cc, "-o", str(exe), str(c)], check=True)
    out = dict(line.split() for line in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    for cname, ct in pairs.items():
        assert int(out[cname]) == ctypes.sizeof(ct), (cname, out[cname], ctypes.sizeof(ct))
    for cname, field in last.items():
        assert int(out[f"{cname}.{field}"]) == getattr(pairs[cname], field).offset, (cname, field)


def test_argument_validation_without_gpu():
    """Validation happens on the host before any launch, so it is testable without a GPU."""
    lib = N.load()
    g = N.GemmArgs()
    assert lib.sm_gemm_f32(g, None) == -1 and b"null pointer" in lib.sm_last_error()
    assert lib.sm_layernorm_f32(None, 384, None, None, None, 384, 4, 384, 1e-6, None) == -1
    assert lib.sm_layernorm_rows_f32(N.LnArgs(), None) == -1
    w = N.Weights()
    w.patch = 7
    assert lib.sm_forward_workspace_bytes(w, 1, 224, 224) == 0
    w.patch, w.n_dec_layers, w.n_queries, w.pos_grid = 16, 6, 20, 14
    b1, b64 = lib.sm_forward_workspace_bytes(w, 1, 224, 224), lib.sm_forward_workspace_bytes(w, 64, 224, 224)
    assert 0 < b1 < b64 < 2 ** 31


@pytest.mark.parametrize("patch,ubc", [(16, True), (8, True), (16, False)])
def test_state_dict_contract(patch, ubc):
    m = MaskFormer(n_queries=20, patch_size=patch, n_decoder_layers=6, return_intermediate=ubc,
                   use_binary_classifier=ubc)
    exp = state_shapes(20, patch, 6, ubc)
    sd = m.state_dict()
    assert list(sd.keys()) == list(exp.keys())
    assert all(tuple(v.shape) == exp[k] for k, v in sd.items())
    m.load_state_dict(synthetic_state_dict(0, "soft", patch_size=patch, use_binary_classifier=ubc), strict=True)
    # wrapped checkpoints ({'model': sd}) are what app.py:185-186 loads
    assert m.encoder.n_embs == 384 and m.encoder.n_heads == 6 and m.encoder.depth == 12 and m.encoder.patch_size == patch


def test_get_model_reads_reference_config_keys(tmp_path):
    from argparse import Namespace
    cfg = Namespace(n_queries=20, n_decoder_layers=6, learnable_pixel_decoder=False, lateral_connection=False,
                    loss_every_decoder_layer=True, scale_factor=2, abs_2d_pe_init=False, use_binary_classifier=True,
                    arch="vit_small", training_method="dino", patch_size=8)
    m = get_model("maskformer", configs=cfg)
    assert isinstance(m, MaskFormer) and m.use_binary_classifier and m.encoder.patch_size == 8
    with pytest.raises(ValueError):
        get_model("resnet50", training_method="swav")
    # both checkpoint wrappers
    from selfmask_amd import load_checkpoint
    sd = synthetic_state_dict(0, "soft", patch_size=8)
    torch.save(sd, tmp_path / "raw.pt")
    torch.save({"model": sd, "n_epochs": 12}, tmp_path / "wrapped.pt")
    load_checkpoint(m, str(tmp_path / "raw.pt"))
    load_checkpoint(m, str(tmp_path / "wrapped.pt"))


def test_no_cpu_fallback():
    m = MaskFormer(n_queries=20, patch_size=16, n_decoder_layers=6, return_intermediate=True, use_binary_classifier=True)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(1, 3, 224, 224))
    bs = BaseStructure(m, device=torch.device("cpu"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        bs._forward({"x": torch.zeros(1, 3, 32, 32)}, device=torch.device("cpu"))
