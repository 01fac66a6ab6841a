"""grayskull_amd -- MI355X-native Grayskull hot path (binding filled in below)."""
