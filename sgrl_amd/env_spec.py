"""Per-environment task constants (reward / termination / reset noise / target sampling).

The reference keeps one Python file per morphology (reference src/environments/<name>.py; 58 files,
20 distinct contents) whose only differences are the constants below (SURVEY.md 8 a3'):

  walker   done = not(lo < h < hi and |pitch| < 1 and |roll| < 1)   3d_walker_7_full.py:33-38
  humanoid same rule, lo/hi shifted by 0.165375                      3d_humanoid_9_full.py:35
  hopper   done = not(finite(s) and |s[3:]| < 100 and h > lo and |torso_ang| < 1), no heading
           reward term                                                3d_hopper_3_shin.py:29-42
  cheetah  no alive bonus; h = min(h, z of *_fthigh); done = not(h > .26 and |pitch|,|roll| < 1.35
           and sum(qvel^2) > 1); reset noise U(-.1,.1) / 0.1*N(0,1)   3d_cheetah_14_full.py:29-37,157-159
  *_v2_*   target radius U(10,20) around the current position         3d_walker_v2_7_full.py:44-45,165-166

Verified against tests/golden/env_arith.npz, which was produced by executing those files.
"""
import re

DONE_WALKER, DONE_HOPPER, DONE_CHEETAH = 0, 1, 2

_WALKER_SHIFT = {
    "2_right_leg_left_knee": (0.26, 0.26),
    "3_left_knee_right_knee": (0.26, 0.26),
    "3_left_leg_right_foot": (0.136, 0.0),
    "4_right_knee_left_foot": (0.136, 0.136),
    "5_foot": (0.0, 0.0),
    "5_left_knee": (0.136, 0.0),
    "6_right_foot": (0.0, 0.0),
    "7_full": (0.0, 0.0),
}
_HOPPER_LO = {"3_shin": 0.45, "4_lower_shin": 0.6, "5_full": 0.95}


class EnvSpec(object):
    __slots__ = ("family", "done_rule", "height_lo", "height_hi", "ang_limit", "alive_bonus", "heading_weight",
                 "ctrl_cost", "reset_pos_noise", "reset_vel_noise", "reset_vel_normal", "target_v2", "frame_skip",
                 "height_bodies")

    def as_dict(self):
        return {k: getattr(self, k) for k in self.__slots__}


def env_spec_for(name):
    """`name` is the environment / XML base name, e.g. '3d_walker_7_full' or '3d_walker_v2_7_full'."""
    m = re.match(r"3d_(walker|hopper|humanoid|cheetah)_(v2_)?(.*)$", name)
    if not m:
        raise KeyError("unknown environment family for %r" % name)
    fam, v2, variant = m.group(1), bool(m.group(2)), m.group(3)
    s = EnvSpec()
    s.family = fam
    s.target_v2 = v2
    s.frame_skip = 4
    s.ctrl_cost = 1e-3
    s.alive_bonus = 1.0
    s.heading_weight = 1.0
    s.ang_limit = 1.0
    s.reset_pos_noise = 0.005
    s.reset_vel_noise = 0.005
    s.reset_vel_normal = False
    s.height_bodies = []
    s.height_hi = 1e30
    if fam == "walker":
        lo, hi = _WALKER_SHIFT[variant]
        s.done_rule = DONE_WALKER
        s.height_lo = 0.8 - lo
        s.height_hi = 2.0 - hi
    elif fam == "humanoid":
        s.done_rule = DONE_WALKER
        s.height_lo = 1.0 - 0.165375
        s.height_hi = 2.0 - 0.165375
    elif fam == "hopper":
        s.done_rule = DONE_HOPPER
        s.height_lo = _HOPPER_LO[variant]
        s.heading_weight = 0.0
    else:
        s.done_rule = DONE_CHEETAH
        s.height_lo = 0.26
        s.ang_limit = 1.35
        s.alive_bonus = 0.0
        s.reset_pos_noise = 0.1
        s.reset_vel_noise = 0.1
        s.reset_vel_normal = True
        s.height_bodies = ["right_fthigh", "left_fthigh"]
    return s
