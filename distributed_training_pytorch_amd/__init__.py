"""distributed_training_pytorch_amd — an MI355X-native (gfx950 / CDNA4) data-parallel training
framework with the capabilities of ducphuongbk01/Distributed-Training-Pytorch.

Layers (see SURVEY.md §1 for the reference's layer map this mirrors):
  utils/     Logger, throughput meter, profiler ranges, checkpoint I/O
  parallel/  rank env, process-group lifecycle, sharded sampler, bucketed gradient reducer,
             DataParallel wrapper, native RCCL transport
  ops/       hand-written HIP kernels (csrc/) behind autograd Functions / nn.Modules
  models/    VGG16 / VGG16-BN on those ops, flat parameter arena
  data/      synthetic datasets, PIL image-folder dataset, async pinned prefetcher
  trainer/   template-method Trainer + the worked ExampleTrainer
"""
__version__ = "0.1.0"
