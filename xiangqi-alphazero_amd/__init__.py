"""xiangqi-alphazero_amd -- MI355X-native batched Xiangqi self-play path.

Holds only what the self-play hot path needs:
  csrc/        hand-written HIP kernels (gfx950) + the C-ABI library declared in include/xq_hip.h
  hip.py       ctypes loader of that library (fails loudly when it is missing)
  engine.py    device-resident self-play engine (select -> evaluate -> expand/backup)
  game_core.py drop-in for the reference's Cython `game_core` plug point (cy_* functions)
  model.py     policy/value ResNet with the reference's state_dict keys
  selfplay.py  `parallel_self_play(model, config, ...)` with the reference's return schema
  weights.py   deterministic counter-based weights (no checkpoints ship with the reference)

The directory name carries a hyphen (project naming); import it as `xiangqi_alphazero_amd`
(the one-file alias module at the repository root).
"""
__version__ = "0.1.0"
