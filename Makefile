# convenience targets (the driver uses __graft_entry__.py / pytest / bench.py directly)
.PHONY: build test-cpu test-gpu bench profiles golden
build:
	python -c "import __graft_entry__ as g; g.build()"
test-cpu: build
	python -m pytest tests -q -m "not gpu"
test-gpu: build
	python -m pytest tests -q -m gpu
bench: build
	python bench.py
profiles:            # on an MI355X box; then copy gpurun_out/r2_* into profiles/
	bash tools/collect_profiles.sh r2
golden:              # needs /root/reference and PIL (build container only)
	python tools/gen_golden.py
