from .helpers import class_to_dict, get_args, set_seed, make_env, merge_dict  # noqa: F401
