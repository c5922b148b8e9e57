"""Static variables of xt/algorithm/impala/default_config.py."""
GAMMA = 0.99
BATCH_SIZE = 512
