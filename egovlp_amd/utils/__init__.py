from .util import load_checkpoint_file, state_dict_data_parallel_fix  # noqa: F401
