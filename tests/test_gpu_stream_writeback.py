"""-m gpu: BASELINE configs[4] whole — `bench.py --stream T --write-back {disk,noop}`: per wave the map/reduce, IssuerMetadata
first sightings, PEM of the NEW list on the GPU and the storage backend (LocalDiskBackend through pinned host buffers and the
native host writer, double-buffered; or NoopBackend).  The run checks itself: every wave's duplicate structure against the
generator, files handed to the backend = NEW certificates, and wave 0's directory tree against the oracle's NEW set —
same paths, same bytes, dirty markers."""
import pytest

pytestmark = pytest.mark.gpu

from tests.test_gpu_bench_multirank import run_bench  # noqa: E402


def test_stream_with_local_disk_write_back(tmp_path):
    d = run_bench(["--stream", "300000", "--entries", "100000", "--write-back", "disk", "--write-back-dir", str(tmp_path),
                   "--write-back-threads", "4", "--traffic", "off"])
    wb = d["write_back"]
    assert wb["backend"] == "LocalDiskBackend" and wb["ok"] is True
    assert wb["files_handed_to_the_backend"] == wb["new_certificates"] == d["result"]["n_new"] > 200000
    assert wb["long_serials_left_to_the_host_parse"] == 0
    c = wb["check_wave0_vs_oracle"]
    assert c["paths_equal_the_oracles_new_set"] and c["files_differing"] == 0 and c["files_compared_bytewise"] == c["oracle_new"]
    assert c["oracle_new"] == c["gpu_new"] == c["files_on_disk"] and c["dirty_markers_cover_the_sampled_days"]
    assert d["result"]["duplicate_structure_matches_generator_in_every_wave"] is True
    assert wb["host_files_per_s"] > 0 and 0.0 <= wb["stall_fraction_of_the_timed_region"] <= 1.0
    assert wb["meta_first_sightings"] > 0 and wb["pem_bytes"] > 2000 * wb["new_certificates"]
    assert "write-back" in d["config"]["workload"]


def test_stream_with_the_noop_backend():
    d = run_bench(["--stream", "300000", "--entries", "100000", "--write-back", "noop", "--traffic", "off"])
    wb = d["write_back"]
    assert wb["backend"] == "NoopBackend" and wb["ok"] is True
    assert wb["files_handed_to_the_backend"] == wb["new_certificates"] == d["result"]["n_new"]
    assert d["result"]["duplicate_structure_matches_generator_in_every_wave"] is True and wb["pem_GB_per_s_of_the_encode_kernels"] > 0
