"""Mirror of the reference's ``retrieval`` package for stage-1 (Wasserstein) retrieval."""
from .two_stage_retrieval import (LoopClosureCandidate, ShardedTwoStageRetrieval, TwoStageRetrieval,
                                  batch_loop_closing, create_two_stage_retrieval)
from .wasserstein import (WassersteinRetriever, wasserstein_distance_1d_numpy, wasserstein_distance_1d_torch,
                          wasserstein_distance_batch_numpy, wasserstein_distance_batch_torch,
                          wasserstein_distance_matrix_numpy, wasserstein_distance_matrix_torch)

__all__ = ["LoopClosureCandidate", "ShardedTwoStageRetrieval", "TwoStageRetrieval", "batch_loop_closing",
           "create_two_stage_retrieval", "WassersteinRetriever", "wasserstein_distance_1d_numpy", "wasserstein_distance_1d_torch",
           "wasserstein_distance_batch_numpy", "wasserstein_distance_batch_torch",
           "wasserstein_distance_matrix_numpy", "wasserstein_distance_matrix_torch"]
