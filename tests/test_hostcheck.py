"""The device headers compiled for the HOST (g++, SES_DEV -> static inline) against the C oracle, bit for bit.

Pre-flight for the GPU: the math / physics device functions are plain IEEE arithmetic, so a transcription
error between simple-es_amd/csrc/*.h and oracle/*.h shows up here, on CPU, before any kernel is launched.
This is test infrastructure only -- the product never runs these functions on the host."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

from oracle import c_oracle as co

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "simple-es_amd", "csrc")

HARNESS = r'''
#include <cmath>
#include "ses_math.h"
#include "ses_cartpole.h"
static ses::TanhEntry TAB[SES_TANH_N];
static void init() { static bool d = false; if (d) return;
  for (int i = 0; i < SES_TANH_N; i++) TAB[i] = {SES_TANH_TABLE[i][0], SES_TANH_TABLE[i][1], SES_TANH_TABLE[i][2], SES_TANH_TABLE[i][3]}; d = true; }
extern "C" {
void v_tanh(const float* x, float* y, int n) { init(); for (int i = 0; i < n; i++) y[i] = ses::tanh_(TAB, x[i]); }
void v_sig(const float* x, float* y, int n) { init(); for (int i = 0; i < n; i++) y[i] = ses::sigmoid_(TAB, x[i]); }
void v_log(const float* x, float* y, int n) { for (int i = 0; i < n; i++) y[i] = ses::log_(x[i]); }
void v_sin(const float* x, float* y, int n) { float c; for (int i = 0; i < n; i++) ses::sincos_(x[i], y[i], c); }
void v_cos(const float* x, float* y, int n) { float s; for (int i = 0; i < n; i++) ses::sincos_(x[i], s, y[i]); }
void v_sin_small(const float* x, float* y, int n) { float c; for (int i = 0; i < n; i++) ses::sincos_small_(x[i], y[i], c); }
void v_cos_small(const float* x, float* y, int n) { float s; for (int i = 0; i < n; i++) ses::sincos_small_(x[i], s, y[i]); }
void v_cp(float* st, const int* a, int* term, int n) { for (int i = 0; i < n; i++) {
  ses::CartPoleState s{st[4*i], st[4*i+1], st[4*i+2], st[4*i+3]}; term[i] = ses::cartpole_step(s, a[i]);
  st[4*i] = s.x; st[4*i+1] = s.xd; st[4*i+2] = s.th; st[4*i+3] = s.thd; } }
}
'''
ORACLE = r'''
#include "ses_oracle_math.h"
void v_tanh(const float*x,float*y,int n){for(int i=0;i<n;i++)y[i]=o_tanhf(x[i]);}
void v_sig(const float*x,float*y,int n){for(int i=0;i<n;i++)y[i]=o_sigmoidf(x[i]);}
void v_log(const float*x,float*y,int n){for(int i=0;i<n;i++)y[i]=o_logf(x[i]);}
void v_sin(const float*x,float*y,int n){float c;for(int i=0;i<n;i++)o_sincosf(x[i],y+i,&c);}
void v_cos(const float*x,float*y,int n){float s;for(int i=0;i<n;i++)o_sincosf(x[i],&s,y+i);}
'''


@pytest.fixture(scope="module")
def libs(tmp_path_factory):
    d = tmp_path_factory.mktemp("hostcheck")
    (d / "h.cpp").write_text(HARNESS)
    (d / "o.c").write_text(ORACLE)
    subprocess.check_call(["g++", "-O2", "-mfma", "-ffp-contract=off", "-std=c++20", "-shared", "-fPIC", "-I", CSRC,
                           str(d / "h.cpp"), "-o", str(d / "h.so")])
    subprocess.check_call(["gcc", "-O2", "-mfma", "-ffp-contract=off", "-shared", "-fPIC", "-I",
                           os.path.join(ROOT, "oracle"), str(d / "o.c"), "-o", str(d / "o.so"), "-lm"])
    return ctypes.CDLL(str(d / "h.so")), ctypes.CDLL(str(d / "o.so"))


def call(lib, fn, x):
    y = np.empty_like(x)
    getattr(lib, fn)(x.ctypes.data_as(ctypes.c_void_p), y.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(x.size))
    return y


def test_device_math_equals_oracle_math_bitwise(libs):
    dev, ora = libs
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.uniform(-100, 100, 500_000), rng.normal(0, 1, 500_000), rng.normal(0, 1e-3, 50_000),
                        [0, np.inf, -np.inf, np.nan, 1e30, -1e30, 10.0, -10.0]]).astype(np.float32)
    for fn in ("v_tanh", "v_sig", "v_sin", "v_cos"):
        a, b = call(dev, fn, x), call(ora, fn, x)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), fn
    xl = np.abs(x[np.isfinite(x) & (x != 0)])
    assert np.array_equal(call(dev, "v_log", xl).view(np.uint32), call(ora, "v_log", xl).view(np.uint32))


def test_small_angle_sincos_is_the_general_one_inside_its_range(libs):
    """sincos_small_ (no argument reduction) == sincos_ == oracle for every |x| <= SINCOS_SMALL_MAX, which is where
    cartpole_pre uses it; checked on all floats of a dense grid plus the range ends."""
    dev, ora = libs
    rng = np.random.default_rng(2)
    x = np.concatenate([rng.uniform(-0.78, 0.78, 2_000_000), rng.normal(0, 0.05, 500_000), rng.normal(0, 1e-6, 10_000),
                        [0.78, -0.78, 0.75, -0.75, 0.20943951, 0.0, 1e-30, -1e-30]]).astype(np.float32)
    x = x[np.abs(x) <= np.float32(0.78)]
    x = x[x.view(np.uint32) != 0x80000000]          # -0.0: sin differs in the sign of zero only (documented)
    for small, full in (("v_sin_small", "v_sin"), ("v_cos_small", "v_cos")):
        a = call(dev, small, x)
        assert np.array_equal(a.view(np.uint32), call(dev, full, x).view(np.uint32)), small
        assert np.array_equal(a.view(np.uint32), call(ora, full, x).view(np.uint32)), small


def test_device_cartpole_equals_oracle_bitwise(libs):
    dev, _ = libs
    rng = np.random.default_rng(1)
    n = 300_000
    st = np.stack([rng.uniform(-3, 3, n), rng.uniform(-3, 3, n), rng.uniform(-0.5, 0.5, n), rng.uniform(-3, 3, n)],
                  1).astype(np.float32)
    act = rng.integers(0, 2, n).astype(np.int32)
    term = np.zeros(n, np.int32)
    got = st.copy()
    dev.v_cp(got.ctypes.data_as(ctypes.c_void_p), act.ctypes.data_as(ctypes.c_void_p),
             term.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(n))
    x, xd, th, thd = [np.ascontiguousarray(st[:, k]) for k in range(4)]
    ret, status = np.zeros(n, np.float32), np.zeros(n, np.uint32)
    co.cartpole_step_soa(0, 0, x, xd, th, thd, act, ret, status)
    assert np.array_equal(np.stack([x, xd, th, thd], 1).view(np.uint32), got.view(np.uint32))
    assert np.array_equal((status >> 31).astype(np.int32), term)


def test_tanh_tables_are_the_same_data():
    def payload(path):
        text = open(path).read()
        return re.findall(r"\{([^{}]+)\},", text), re.findall(r"#define (SES_TANH_\w+) (.+)", text)
    a = payload(os.path.join(CSRC, "ses_tanh_table.h"))
    b = payload(os.path.join(ROOT, "oracle", "ses_tanh_table.h"))
    assert a[0] == b[0] and len(a[0]) == 320
    assert [d for d in a[1] if d[0] != "SES_TANH_TABLE_QUAL"] == b[1]
