"""egovlp_amd -- the EgoClip pre-training step of showlab/EgoVLP on MI355X."""
import os

# The step runs on several HIP streams (the text tower under the video tower, optionally wgrad / optimizer streams, and in
# data-parallel runs RCCL's own).  The HIP runtime multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4):
# once a process group exists, the text tower's stream ends up sharing a queue with the main stream and the overlap of the
# two towers -- 1.4 ms of a 40 ms step -- silently disappears (profiles/r02_d_dp_overhead.txt).  Read by the runtime when it
# initialises, i.e. at the first HIP call: import this package (or set the variable) before touching torch.cuda.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
