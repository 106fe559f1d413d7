from .factory import Factory, KwargsDict
from .ppo import PPOFactory
from .a2c import A2CFactory
