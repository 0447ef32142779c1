"""CPU: the C-ABI library builds (hipcc cross-compiles gfx950 without a GPU), loads, and exports every symbol
include/mvae_hip.h declares; host-only entry points behave (plan/tables/validation); no compute call is made."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from tests.common import CONFIGS, ROOT, engine_args, oracle_config


def _header_functions():
    src = open(os.path.join(ROOT, "include", "mvae_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mvae_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported_and_bound(hip_lib):
    from multiscale_variational_autoencoder_amd import _abi
    declared = _header_functions()
    assert len(declared) >= 20
    assert sorted(_abi.SYMBOLS) == declared            # the ctypes table mirrors the header exactly
    out = subprocess.run(["nm", "-D", "--defined-only", _abi.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r"\bT (mvae_[a-z0-9_]+)", out))
    assert set(declared) <= exported
    for name in declared:
        assert getattr(hip_lib, name) is not None
    assert hip_lib.mvae_abi_version() == _abi.MVAE_ABI_VERSION


@pytest.mark.parametrize("name", sorted(CONFIGS))
def test_plan_tables_match_oracle(hip_lib, name):
    """mvae_create builds the same variable inventory (names, shapes, regularisers, order) as the oracle's
    restatement of _build_encoder/_build_decoder/basic_block."""
    from oracle.mvae_oracle import param_table
    from multiscale_variational_autoencoder_amd.engine import Engine
    eng = Engine(**engine_args(name, 4))
    P, S = param_table(oracle_config(name))
    assert list(eng.param_table) == list(P)
    for k in P:
        assert eng.param_table[k]["shape"] == tuple(P[k][0]) and eng.param_table[k]["reg"] == P[k][1], k
    assert list(eng.state_table) == list(S)
    # arena order: the big Dense weights first (the region [0, reduce_split) the DP path all-reduces early), then the
    # rest in table order
    offs = sorted((v["offset"], int(np.prod(v["shape"]))) for v in eng.param_table.values())
    for (o0, n0), (o1, _) in zip(offs, offs[1:]):
        assert o1 >= o0 + n0 and o1 % 64 == 0          # disjoint, 256-byte aligned arena slots
    D = eng.reduce_split
    assert 0 <= D <= eng.P and all(o + n <= D or o >= D for o, n in offs)
    lead = {k for k, v in eng.param_table.items() if v["offset"] < D}
    assert all(k.endswith((".mu.w", ".log_var.w", ".dense.w")) for k in lead)
    assert eng.Z == sum(oracle_config(name).z_dims)
    assert eng.R >= eng.P + eng.S + 4 + eng.levels
    big = Engine(**engine_args(name, 8))
    assert big.ws_bytes > eng.ws_bytes                  # workspace grows with max_batch
    eng.close(); big.close()


def test_create_validation_errors(hip_lib):
    from multiscale_variational_autoencoder_amd.engine import Engine
    base = engine_args("tiny", 2)
    with pytest.raises(ValueError, match="z_dims elements should be > 0"):
        Engine(**dict(base, z_dims=[4, 0]))
    with pytest.raises(ValueError, match="divisible"):
        Engine(**dict(base, input_dims=(6, 8, 3), z_dims=[4, 4, 4]))
    with pytest.raises(ValueError, match="levels"):
        Engine(**dict(base, z_dims=[4]))                # the reference's merge model needs >= 2 levels
    with pytest.raises(ValueError, match="decoder of scale"):
        Engine(**dict(base, decoder={"filters": [8], "kernel_size": [(3, 3)], "strides": [(1, 1)]}))
    with pytest.raises(ValueError, match="Filters"):
        Engine(**dict(base, encoder={"filters": [0], "kernel_size": [(3, 3)], "strides": [(1, 1)]}))
    with pytest.raises(ValueError, match="max_batch"):
        Engine(**dict(base, max_batch=0))


def test_unbound_handle_refuses_compute(hip_lib):
    """Call-order errors come back as codes + text, never as a crash; nothing runs without mvae_bind."""
    from multiscale_variational_autoencoder_amd import _abi
    from multiscale_variational_autoencoder_amd.engine import Engine
    eng = Engine(**engine_args("tiny", 2))
    io = _abi.MvaeStepIO()
    io.batch, io.training = 2, 1
    assert hip_lib.mvae_forward(eng.h, C.byref(io), None) == _abi.MVAE_E_STATE
    assert b"mvae_bind" in hip_lib.mvae_last_error(eng.h)
    assert hip_lib.mvae_backward(eng.h, 1.0, 1.0, None) == _abi.MVAE_E_STATE
    assert hip_lib.mvae_apply_adagrad(eng.h, 1e-3, 1.0, 1.0, None) == _abi.MVAE_E_STATE
    assert hip_lib.mvae_bind(eng.h, 0, None, None, None, None, None, 0) == _abi.MVAE_E_INVALID
    # the RCCL entry points: no communicator without a bound device, no exchange without a communicator
    ident = C.create_string_buffer(_abi.MVAE_COMM_ID_BYTES)
    assert hip_lib.mvae_comm_init(eng.h, ident, 0, 1) == _abi.MVAE_E_STATE
    assert hip_lib.mvae_comm_size(eng.h) == 0
    assert hip_lib.mvae_allreduce(eng.h, 0, -1, None) == _abi.MVAE_E_STATE
    assert hip_lib.mvae_train_step_dp(eng.h, C.byref(io), 1.0, 1.0, 1e-3, 1.0, None) == _abi.MVAE_E_STATE
    assert hip_lib.mvae_comm_destroy(eng.h) == _abi.MVAE_OK
    p, n = C.c_void_p(), C.c_int64()
    assert hip_lib.mvae_tensor_lookup(eng.h, b"enc0.b0.mn.t1", C.byref(p), C.byref(n)) == _abi.MVAE_OK
    assert p.value is None and n.value == 4 * 4 * 8
    assert hip_lib.mvae_tensor_lookup(eng.h, b"no.such.tensor", C.byref(p), C.byref(n)) == _abi.MVAE_E_INVALID
    eng.close()


def test_missing_library_fails_loudly(tmp_path):
    from multiscale_variational_autoencoder_amd import _abi
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _abi.load_library(str(tmp_path / "libmvae_hip.so"))


def test_release_build_has_no_debug_switches(hip_lib):
    """MVAE_DEBUG_ONLY_SCALE (a work-skipping timing diagnostic) exists only in -DMVAE_DEBUG_BUILD libraries."""
    assert hip_lib.mvae_debug_build() == 0
    out = subprocess.run(["strings", "-a", os.path.join(ROOT, "multiscale_variational_autoencoder_amd", "libmvae_hip.so")],
                         capture_output=True, text=True, check=True).stdout
    assert "MVAE_DEBUG_ONLY_SCALE" not in out


def test_bf16_scale_decision_mirrors_the_launchers(hip_lib):
    """A scale runs bfloat16 only when every bf16 launcher covers its shapes (scale_bf16_ok admitted 5x5
    stride-1 layers and tensors >= 2 GiB, which launch16_taps / launch16_wgrad then refused at run time)."""
    from multiscale_variational_autoencoder_amd.engine import Engine
    nb = {"filters": [64, 64, 32], "kernel_size": [(5, 5), (3, 3), (1, 1)], "strides": [(2, 2), (1, 1), (1, 1)]}
    eng = Engine((64, 64, 3), [8, 8], nb, nb, 0.0, 255.0, 0.01, 4, act_dtype="bf16")
    assert eng.scale_dtypes() == ["bf16", "bf16"]
    eng.close()
    # 5x5 at stride 1: ceil(5/1)^2 = 25 taps per sub-pixel phase > 9 -> T-form has no kernel -> the scale stays float32
    s1 = {"filters": [64, 64, 32], "kernel_size": [(5, 5), (3, 3), (1, 1)], "strides": [(1, 1), (1, 1), (1, 1)]}
    eng = Engine((64, 64, 3), [8, 8], s1, s1, 0.0, 255.0, 0.01, 4, act_dtype="bf16")
    assert eng.scale_dtypes() == ["f32", "f32"]
    eng.close()
    # kernel rows: 6..8 x 5 exceed the 25 taps the launchers stage
    k7 = {"filters": [64, 64, 32], "kernel_size": [(7, 5), (3, 3), (1, 1)], "strides": [(2, 2), (1, 1), (1, 1)]}
    eng = Engine((64, 64, 3), [8, 8], k7, k7, 0.0, 255.0, 0.01, 4, act_dtype="bf16")
    assert eng.scale_dtypes() == ["f32", "f32"]
    eng.close()
    # 32-bit byte offsets: 256x256x64 bf16 at max_batch 256 is 2 GiB -> scale 0 float32, the smaller scales bfloat16
    big = Engine((256, 256, 3), [16] * 7, nb, nb, 0.0, 255.0, 0.01, 256, act_dtype="bf16")
    assert big.scale_dtypes()[0] == "f32" and big.scale_dtypes()[1] == "bf16"
    big.close()
    ok = Engine((256, 256, 3), [16] * 7, nb, nb, 0.0, 255.0, 0.01, 64, act_dtype="bf16")
    assert ok.scale_dtypes()[0] == "bf16"
    ok.close()
