from .env_runner import EnvRunner, RunnerWrapper
from .onpolicy import (TransformInteractions, IterateWithMinibatches, ppo_runner_wrap,
                       make_ppo_runner)
from .summary import PeriodicSummaries
from .trajectory_transforms import GAE, MergeTimeBatch, NormalizeAdvantages, Take
