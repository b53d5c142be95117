#!/usr/bin/env python3
"""Where a BipedalWalker (or LunarLander MLP) rollout spends its wave-cycles, by phase of the env step.  Needs a library
built with -DSES_PHASE_TIMERS (SES_OUT=ab/libT.so csrc/build.sh -DSES_PHASE_TIMERS; SES_LIB_PATH=ab/libT.so): the marks
of ses_b2.h (B2_PHASE) sum, over all waves, the cycles between consecutive marks.
usage: walker_phases.py <lander|walker|c3> [offspring] [lpe:epw]      (c3: LunarLander POMDP GRU, the lockstep kernel)"""
import ctypes, json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "simple-es_amd"))
from ses import HipES
from ses import _lib

which = sys.argv[1] if len(sys.argv) > 1 else "walker"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
lpe, epw = (int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "0:0").split(":"))
env, S = ("BipedalWalker-v3", 24) if which == "walker" else ("LunarLanderContinuous-v2", 8)
NAMES = ["call of the step, state copy in, motors", "collide", "integrate v, pack, contact / joint init, warm start",
         "velocity iterations", "integrate positions, re-pack", "position iterations", "sleep, write-back", "between the halves",
         "time of impact: the rest", "lidar rays (walker)", "reward, state copy out, return, loop", "observation + policy", "call of bw_step (walker)", "state copy in (walker)", "time of impact: scans (reach tests, b2TimeOfImpact)",
         "time of impact: events (contact update, sub-step solve)"]
es = HipES(env, S, 4, False, which == "c3", pomdp=which == "c3", max_step=300, eval_ep_num=5)
es.set_tuning("box2d_lanes_per_env", lpe); es.set_tuning("box2d_envs_per_wave", epw)
theta = es.perturb(es.zeros(es.P), 0.5 if which == "c3" else 2.0, 0, 0, 0, n)
init = es.init_states_uniform(0, 0, 0, n)
fit = es.empty(n)
lib = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_ulonglong * 24)()
es.rollout(theta, init, fitness=fit); torch.cuda.synchronize()
assert lib.ses_debug_phase_totals(buf, 1) == 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); es.rollout(theta, init, fitness=fit); e1.record(); e1.synchronize()
assert lib.ses_debug_phase_totals(buf, 1) == 0
tot = sum(buf[:17])
out = {"env": env, "offspring": n, "lanes_per_env": lpe, "envs_per_wave": epw, "rollout_ms_with_timers": round(e0.elapsed_time(e1), 2),
       "share": {NAMES[k]: round(buf[k] / tot, 4) for k in range(16)}}
if buf[16]:
    # library built with -DSES_PHASE_SPLIT_VEL as well: the velocity iterations' joints ("velocity iterations" above) and contact rows apart
    out["share"]["velocity iterations: joints only"] = out["share"].pop("velocity iterations")
    out["share"]["velocity iterations: contact rows"] = round(buf[16] / tot, 4)
if buf[19]:
    # census over the world steps that had a contact: contact-row slots a wave executes per velocity iteration today (the union over its
    # envs, body by body) and what a flat per-env list would execute (the largest total of any one env)
    out["contact_rows_per_iteration"] = {"per_body_union": round(buf[17] / buf[19], 3), "flat_per_env_list": round(buf[18] / buf[19], 3),
                                         "flat_over_union": round(buf[18] / buf[17], 3)}
print(json.dumps(out, indent=1))
