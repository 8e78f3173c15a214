# Top-level conveniences.  `make` builds everything __graft_entry__.build() builds; `make asan` builds the host-side code
# (the oracle, the host half of the library + the test emulator of the kernel bodies, the command line) with
# AddressSanitizer + UndefinedBehaviorSanitizer and runs the CPU test-suite against those builds.
# (GPU sanitizers are not available on this pool; the device code is covered by running the same kernel bodies in the
# sanitized emulator.)
PY ?= python3
ASAN_DIR := build/asan
SAN := -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -O1 -g
LIBASAN := $(shell gcc -print-file-name=libasan.so)

.PHONY: all asan asan-build clean
all:
	$(PY) -c "import __graft_entry__ as g; g.build()"

asan-build:
	mkdir -p $(ASAN_DIR)
	gcc -std=gnu11 $(SAN) -fPIC -shared -Wall -Wextra -Wno-unused-parameter -o $(ASAN_DIR)/liboracle.so oracle/dbtk_oracle.c -ldl -lm
	g++ -std=c++17 $(SAN) -fPIC -shared -Wall -Wno-unused-function -Wno-unknown-pragmas -DDBTK_LOC_Q=40 -o $(ASAN_DIR)/libdbtk_emu.so tests/emu/emu.cpp danbing-tk_amd/csrc/dbtk_rpgg.cpp -lpthread
	g++ -std=c++17 $(SAN) -Wall -o $(ASAN_DIR)/ktools danbing-tk_amd/csrc/dbtk_ktools.cpp danbing-tk_amd/csrc/dbtk_rpgg.cpp -lpthread
	g++ -std=c++17 $(SAN) -Wall -o $(ASAN_DIR)/danbing-tk danbing-tk_amd/csrc/dbtk_cli.cpp -Ldanbing-tk_amd -ldbtk_hip -Wl,-rpath,'$$ORIGIN/../../danbing-tk_amd' -lpthread -lz

# DBTK_NO_REF: the tests that drive the compiled reference (oracle/_ref) are skipped — it aborts under ASan's allocator and is
# not ours to fix; RPGG files then come from tests/synth.py's own builder.
# detect_leaks=0: the Python interpreter itself is not leak-clean.  The emulator switches stacks by hand (coroutine lanes),
# which ASan's stack-use-after-return bookkeeping cannot follow: that one check is off, everything else is on.
asan: asan-build
	LD_PRELOAD=$(LIBASAN) ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1 \
	DBTK_NO_REF=1 DBTK_ASAN=1 DBTK_TIDY_EXIT=1 DBTK_ORACLE_LIB=$(CURDIR)/$(ASAN_DIR)/liboracle.so DBTK_EMU_LIB=$(CURDIR)/$(ASAN_DIR)/libdbtk_emu.so DBTK_CLI=$(CURDIR)/$(ASAN_DIR)/danbing-tk \
	$(PY) -m pytest tests -x -q -m "not gpu" -k "not distributed" -p no:cacheprovider

clean:
	rm -rf build
