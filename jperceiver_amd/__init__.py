"""jperceiver_amd — MI355X-native JPerceiver `Baseline` training step (hand-written HIP kernels behind the
reference's module / loss / trainer API).  Importing the package does not need a GPU; running it does."""
__version__ = "0.1.0"
