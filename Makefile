# Top-level Makefile: the reference's build contract (Makefile:17-21 there, run.sh:1-2):
#   make            -> ./simulation.out   (runs with no arguments, writes ./data/*.bin)
#   make clean / make rebuild
# plus `make lib`, `make oracle`, `make test`.
HIPCC ?= /opt/rocm/bin/hipcc
TARGET = simulation.out
LIBDIR = fluid_simulation_amd
LIB = $(LIBDIR)/libfluidsim.so

all: $(TARGET)

lib:
	$(MAKE) -C $(LIBDIR)/csrc

$(LIB): lib

$(TARGET): src/main.cpp include/fluidsim.h $(LIB)
	$(HIPCC) -O2 -std=c++17 src/main.cpp -o $@ -L$(LIBDIR) -lfluidsim -Wl,-rpath,'$$ORIGIN/$(LIBDIR)'

oracle:
	$(MAKE) -C oracle

test:
	python -m pytest tests -x -q -m "not gpu"

clean:
	$(RM) $(TARGET)
	$(MAKE) -C $(LIBDIR)/csrc clean

rebuild: clean all

.PHONY: all lib oracle test clean rebuild
