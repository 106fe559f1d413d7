"""derl_amd -- MI355X-native rollout + PPO/A2C update engine behind derl's Python API."""
