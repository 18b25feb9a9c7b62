"""rafft_amd - MI355X-native RAFFT folding engine (drop-in for `rafft.fold`).

    from rafft_amd import fold
    structures, trajectory = fold(seq, max_stack=20, traj=True)

The compute path is libraffthip.so (hand-written HIP for gfx950); there is no CPU
fallback - importing works anywhere, calling needs the built library and a GPU."""
from .rafft import fold, fold_batch, submit_batch, eval_structures, eval_structures_info, last_stats  # noqa: F401
from .params import load_params, load_params_from_viennarna, reset_params, save_params, params_info, unpinned_entries  # noqa: F401
from .rafft_kin import kinetics  # noqa: F401
from .utils import Structure, parse_rafft_output, paired_positions, dot_bracket, read_fasta, format_trajectory  # noqa: F401
