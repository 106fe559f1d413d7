from .common import Alg, Loss, Trainer, r_squared, total_norm
from .ppo import PPO, PPOLoss
from .a2c import A2C, A2CLoss
