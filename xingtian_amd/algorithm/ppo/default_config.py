"""Static variables of xt/algorithm/ppo/default_config.py."""
GAMMA = 0.99
LAM = 0.95
BATCH_SIZE = 512
