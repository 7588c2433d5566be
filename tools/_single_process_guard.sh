# sourced by the profiling scripts: rocprofv3 has initialised the GPU before bench.py starts, and bench.py --gpus N would start
# torchrun from that process (a re-exec from a GPU-initialised process takes the box down on this pool)
case " $* " in *" --gpus "*) echo "$0: single-process runs only (no --gpus under rocprofv3)" >&2; exit 2;; esac
