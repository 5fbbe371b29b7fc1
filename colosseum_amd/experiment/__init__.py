from .csv_logger import CSVLogger  # noqa: F401
from .mdp_loop import InMemoryLogger, MDPLoop, MDPSpec, make_mdp_spec  # noqa: F401
