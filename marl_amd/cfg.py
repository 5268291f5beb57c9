"""Environment and training flags with the names and defaults of the reference's
argparse block (marl_llm/cfg/assembly_cfg.py:159-213), as a plain dataclass.  The
reference module also turns fig/*.png into grids at import time (cfg:35-155); here that
is marl_amd.shapes, and the seven resulting grids ship as marl_amd/data/shapes.npz."""
from dataclasses import dataclass
from typing import Optional


@dataclass
class AssemblyArgs:
    # environment (cfg:159-179)
    n_a: int = 30
    is_boundary: bool = True
    is_con_self_state: bool = True
    is_feature_norm: bool = False
    dynamics_mode: str = "Cartesian"
    render_traj: bool = False           # the reference default is True; nothing renders here
    traj_len: int = 15
    agent_strategy: str = "input"
    training_method: str = "llm_rl"
    is_collected: bool = False
    results_file: Optional[str] = None  # None: packaged grids
    live_render: bool = False
    # training (cfg:184-208)
    env_name: str = "assembly"
    seed: int = 226
    n_rollout_threads: int = 1
    n_training_threads: int = 5
    buffer_length: int = int(5e4)
    n_episodes: int = 3000
    episode_length: int = 200
    batch_size: int = 2048
    hidden_dim: int = 256
    lr_actor: float = 1e-4
    lr_critic: float = 1e-3
    epsilon: float = 0.1
    noise_scale: float = 0.9
    tau: float = 0.01
    agent_alg: str = "MADDPG"
    device: str = "gpu"
    save_interval: int = 10


gpsargs = AssemblyArgs()
