"""Accuracy of the oracle's deterministic fp32 elementary functions vs float64 libm.

The bounds asserted here are the ones quoted in oracle/ses_oracle_math.h."""
import ctypes
import os
import subprocess
import tempfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SRC = r'''
#include "ses_oracle_math.h"
void v_exp(const float*x,float*y,int n){for(int i=0;i<n;i++)y[i]=o_expf(x[i]);}
void v_tanh(const float*x,float*y,int n){for(int i=0;i<n;i++)y[i]=o_tanhf(x[i]);}
void v_sig(const float*x,float*y,int n){for(int i=0;i<n;i++)y[i]=o_sigmoidf(x[i]);}
void v_log(const float*x,float*y,int n){for(int i=0;i<n;i++)y[i]=o_logf(x[i]);}
void v_sin(const float*x,float*y,int n){float c;for(int i=0;i<n;i++)o_sincosf(x[i],y+i,&c);}
void v_cos(const float*x,float*y,int n){float s;for(int i=0;i<n;i++)o_sincosf(x[i],&s,y+i);}
'''


@pytest.fixture(scope="module")
def mlib():
    d = tempfile.mkdtemp()
    c = os.path.join(d, "m.c")
    so = os.path.join(d, "m.so")
    open(c, "w").write(SRC)
    subprocess.check_call(["gcc", "-O2", "-mfma", "-ffp-contract=off", "-shared", "-fPIC",
                           "-I", os.path.join(ROOT, "oracle"), c, "-o", so, "-lm"])
    return ctypes.CDLL(so)


def call(lib, fn, x):
    x = np.ascontiguousarray(x, dtype=np.float32)
    y = np.empty_like(x)
    getattr(lib, fn)(x.ctypes.data_as(ctypes.c_void_p), y.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(x.size))
    return y


def ulp_err(y, ref64):
    sp = np.spacing(np.abs(ref64.astype(np.float32))).astype(np.float64)
    return np.abs(y.astype(np.float64) - ref64) / sp


def test_exp(mlib):
    x = np.random.default_rng(0).uniform(-86, 88, 400_000).astype(np.float32)
    assert ulp_err(call(mlib, "v_exp", x), np.exp(x.astype(np.float64))).max() <= 1.0
    assert call(mlib, "v_exp", np.array([0.0], np.float32))[0] == 1.0
    assert np.isfinite(call(mlib, "v_exp", np.array([-1e9, 1e9, -87.5, 88.5], np.float32))).all()


def test_tanh(mlib):
    """table-driven piecewise cubic (tools/gen_tanh_table.py): absolute error bound, exact oddness,
    exact zero, saturation to +-1."""
    rng = np.random.default_rng(1)
    x = np.concatenate([rng.uniform(-12, 12, 300_000), rng.normal(0, 0.3, 200_000), rng.normal(0, 1e-3, 20_000),
                        np.arange(0, 10.5, 1 / 32), np.nextafter(np.arange(1 / 32, 10.5, 1 / 32, dtype=np.float32), 0),
                        [0.0, 50.0, -50.0, 9.0, 8.9999]]).astype(np.float32)
    y = call(mlib, "v_tanh", x)
    ref = np.tanh(x.astype(np.float64))
    assert np.abs(y - ref).max() <= 1.2e-7
    small = np.abs(x) < 1 / 32
    assert ulp_err(y[small], ref[small]).max() <= 1.5                 # Taylor interval keeps relative accuracy near 0
    assert (np.abs(y) <= 1.0).all() and (np.sign(y) == np.sign(x)).all()
    assert (call(mlib, "v_tanh", -x) == -y).all()                      # exactly odd
    assert call(mlib, "v_tanh", np.array([0.0, 10.0, 1e30, -1e30, np.inf], np.float32)).tolist() == [0.0, 1.0, 1.0, -1.0, 1.0]


def test_sigmoid(mlib):
    x = np.random.default_rng(2).uniform(-30, 30, 300_000).astype(np.float32)
    y = call(mlib, "v_sig", x)
    ref = 1 / (1 + np.exp(-x.astype(np.float64)))
    assert np.abs(y - ref).max() <= 1.2e-7
    assert ((y >= 0) & (y <= 1)).all()


def test_log(mlib):
    rng = np.random.default_rng(3)
    x = np.concatenate([rng.uniform(1e-10, 1, 300_000), 2.0 ** -rng.uniform(0, 33, 50_000), [1.0]]).astype(np.float32)
    y = call(mlib, "v_log", x)
    assert ulp_err(y, np.log(x.astype(np.float64))).max() <= 1.0
    assert y[-1] == 0.0


@pytest.mark.parametrize("R", [0.3, 4.0, 100.0, 8192.0])
def test_sincos(mlib, R):
    x = np.random.default_rng(4).uniform(-R, R, 300_000).astype(np.float32)
    xs = x.astype(np.float64)
    assert np.abs(call(mlib, "v_sin", x) - np.sin(xs)).max() <= 1.0e-7
    assert np.abs(call(mlib, "v_cos", x) - np.cos(xs)).max() <= 1.0e-7
