"""Base reward signal (reference: pypownet/reward_signal.py): a template returning ``[0.]``."""


class RewardSignal(object):
    def __init__(self):
        pass

    def compute_reward(self, observation, action, flag):
        return [0.]
