"""MI355X-native drop-in for the `gym_vrp` package of kevin-schumann/VRP-GYM.

Same import surface (`gym_vrp.envs.{TSPEnv,VRPEnv,IRPEnv}`); state lives in
contiguous device tensors and every step runs in libvrpgym_hip.so.
"""
