"""Environment construction with derl's entry point ``derl.env.make(env_id, nenvs, seed)``
(derl/env/make_env.py:170-185).  The real Atari / MuJoCo wrapper stacks are outside this
build's scope; ids of those families map to device-resident synthetic envs
with the same observation / action contract, and CartPole-v1 is built in."""
from .spaces import Box, Discrete, Space, is_box, is_discrete
from .synthetic import SyntheticAtariEnv, SyntheticMuJoCoEnv
from .cartpole import CartPoleBatch
from .env_batch import EnvBatch, ParallelEnvBatch, SingleEnvBatch, SpaceBatch
from .bridge import HostEnvBridge
from .atari_device import DeviceAtariFrames
from .normalize import Normalize
from .summarize import DeviceSummarize, RewardSummarizer, Summarize

ATARI_ACTIONS = {"Breakout": 4, "SpaceInvaders": 6, "Pong": 6, "BeamRider": 9, "Qbert": 6,
                 "Seaquest": 18, "Enduro": 9}
MUJOCO_DIMS = {"HalfCheetah": (17, 6), "Hopper": (11, 3), "Walker2d": (17, 6), "Ant": (111, 8),
               "Swimmer": (8, 2), "Reacher": (11, 2), "InvertedPendulum": (4, 1),
               "InvertedDoublePendulum": (11, 1), "Humanoid": (376, 17)}


def _base_name(env_id):
  name = env_id[:env_id.rfind("-")] if "-" in env_id else env_id
  for postfix in ("Deterministic", "NoFrameskip", "BulletEnv"):
    if name.endswith(postfix):
      name = name[:-len(postfix)]
  return name


def is_atari_id(env_id):
  """derl/env/make_env.py:48-57 (game list reduced to the table above)."""
  return _base_name(env_id) in ATARI_ACTIONS


def is_mujoco_id(env_id):
  """derl/env/make_env.py:60-66."""
  return _base_name(env_id) in MUJOCO_DIMS


def make(env_id, nenvs=None, seed=0, device="cuda", rank=0, normalize=False, summarize=False,
         **kwargs):
  """Creates a batched env.  nenvs=None means one env (derl's unbatched case maps to a
  batch of 1 here).  ``normalize=True`` puts the device ``Normalize`` wrapper on a
  MuJoCo-family env like derl's mujoco_wrap does (make_env.py:158-167); the synthetic stand-in
  already produces observations in the wrapper's output range, so it is off by default.
  ``summarize=True`` adds the reward summaries of derl's Summarize wrapper (make_env.py:108-109,
  123-124; tags ``<env_id>/total_reward`` ...), advanced on the device once per rollout."""
  del kwargs
  seed = 0 if seed is None else seed
  if env_id.startswith("CartPole"):
    return CartPoleBatch(nenvs or 1, seed)
  if is_atari_id(env_id):
    env = SyntheticAtariEnv(nenvs or 1, ATARI_ACTIONS[_base_name(env_id)], seed,
                            device=device, rank=rank)
    return DeviceSummarize(env, env_id) if summarize else env
  if is_mujoco_id(env_id):
    obs_dim, act_dim = MUJOCO_DIMS[_base_name(env_id)]
    env = SyntheticMuJoCoEnv(nenvs or 1, obs_dim, act_dim, seed, device=device, rank=rank)
    env = DeviceSummarize(env, env_id) if summarize else env  # raw rewards, like mujoco_wrap
    return Normalize(env) if normalize else env
  raise ValueError(f"unknown env id {env_id!r}: this build provides CartPole-v1 and synthetic "
                   f"stand-ins for {sorted(ATARI_ACTIONS)} / {sorted(MUJOCO_DIMS)}")
