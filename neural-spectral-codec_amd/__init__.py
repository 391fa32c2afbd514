"""MI355X-native descriptor hot path of Neural-Spectral-Codec (encoder + GAT enhancer).

Sub-packages mirror the reference's module paths for the path:
  encoding.spectral_encoder.SpectralEncoder      <- src/encoding/spectral_encoder.py:24
  encoding.range_image.RangeImageProjector       <- src/encoding/range_image.py:92
  gnn.model.{SpectralGNN,create_spectral_gnn}    <- src/gnn/model.py:21,284
  keyframe.graph_manager.build_graph_from_keyframes_batch <- src/keyframe/graph_manager.py:471
  encoding.quantization.{HistogramQuantizer,CompressedDescriptor} <- src/encoding/quantization.py:22,113
  data.pose_utils.compute_overlap                <- src/data/pose_utils.py:323
  retrieval.wasserstein, gnn.trainer, gnn.triplet_miner     <- the consumers either side
All compute goes through the C-ABI library csrc/libnsc_hip.so (include/nsc.h); there is no
CPU fallback.
"""
__version__ = "0.1.0"
