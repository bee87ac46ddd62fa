# Launch layer (reference: Makefile:39-52).  `make fit` is the target the reference's help text
# promises but never defines; it runs the engine-backed fit loop.
PYTHON ?= python
PKG := implicit-image-compression_amd
KWARGS ?=

.PHONY: build fit test test-gpu bench clean

## build: compile libsiren_fit.so for gfx950
build:
	$(PYTHON) __graft_entry__.py build

## fit: implicit MLP image fitting, e.g. make fit KWARGS="mlp.hidden_size=256 img.height=1024 img.width=1024"
fit: build
	PYTHONPATH=$(PKG) $(PYTHON) -m implicit_image.fit $(KWARGS)

## fit8: one fit per GPU over a comma sweep (per-image sharding), e.g. KWARGS="img.seed=0,1,2,3,4,5,6,7"
fit8: build
	PYTHONPATH=$(PKG) $(PYTHON) -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
	  -m implicit_image.fit $(KWARGS)

test:
	$(PYTHON) -m pytest tests -x -q -m "not gpu"
test-gpu:
	$(PYTHON) -m pytest tests -x -q -m gpu
bench:
	$(PYTHON) bench.py
clean:
	$(MAKE) -C $(PKG)/csrc clean
