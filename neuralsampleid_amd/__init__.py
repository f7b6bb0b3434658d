"""MI355X-native (gfx950) implementation of NeuralSampleID's GNN contrastive-fingerprint path.

Mirrors the reference's module paths for the hot path only:
    neuralsampleid_amd.encoder.graph_encoder.GraphEncoder      (encoder/graph_encoder.py)
    neuralsampleid_amd.encoder.gcn_lib.torch_vertex.Grapher    (encoder/gcn_lib/torch_vertex.py)
    neuralsampleid_amd.encoder.gcn_lib.torch_edge.DenseDilatedKnnGraph
    neuralsampleid_amd.simclr.simclr.SimCLR, neuralsampleid_amd.simclr.ntxent.ntxent_loss
    neuralsampleid_amd.peak_extractor.GPUPeakExtractorv2
All compute runs in libnsid_hip.so (C ABI: include/nsid.h); importing the ops without that library raises."""
__version__ = "0.1.0"
