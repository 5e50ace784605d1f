# Top-level targets. The product library itself is built by hyper-greco_amd/csrc/Makefile (make -C hyper-greco_amd/csrc).
#   make          the product library and the CPU oracle (what __graft_entry__.build() does)
#   make asan     build/asan/libhypergreco.so: the same sources with AddressSanitizer + UndefinedBehaviorSanitizer on the HOST code only
#                 (-fno-gpu-sanitize: GPU sanitizers are not available on this pool), then the CPU test suite's host-logic / loader /
#                 verifier tests against it                                       -> profiles/r06_asan_tests.txt
#   make fuzz     libFuzzer runs (FUZZ_SECONDS each, default 600) of tests/fuzz/fuzz_host.cpp over the JSON witness loaders and both host
#                 verifiers, against the sanitized library                        -> profiles/r06_fuzz_*.txt
ROOT := $(abspath .)
CLANG := /opt/rocm/lib/llvm/bin/clang++
ASAN_RT := $(shell $(CLANG) --print-file-name=libclang_rt.asan-x86_64.so)
SAN := -fsanitize=address,undefined -fno-gpu-sanitize -fno-omit-frame-pointer -g -shared-libsan
FUZZ_SECONDS ?= 600
# reports go to files (pytest captures stderr, and a halted process would take the report with it): build/asan/report.<pid>
SAN_ENV := LD_PRELOAD=$(ASAN_RT) ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:log_path=$(ROOT)/build/asan/report UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1:log_path=$(ROOT)/build/asan/report

all:
	$(MAKE) -C hyper-greco_amd/csrc
	$(MAKE) -C oracle

build/asan/libhypergreco.so: $(wildcard hyper-greco_amd/csrc/*.hip hyper-greco_amd/csrc/*.cpp hyper-greco_amd/csrc/*.hpp hyper-greco_amd/csrc/*.inc) include/hg.h
	bash scripts/build_variant.sh asan "$(SAN) -fsanitize=fuzzer-no-link"

asan: build/asan/libhypergreco.so
	$(MAKE) -C oracle
	mkdir -p profiles
	rm -f build/asan/report.*
	( echo "# make asan: $(SAN) on the host code of every source; python -m pytest -m 'not gpu' host-logic, loader and verifier tests against build/asan/libhypergreco.so"; \
	  HG_LIB=$(ROOT)/build/asan/libhypergreco.so $(SAN_ENV) python -m pytest tests/test_host_logic.py tests/test_oracle_kats.py -q -m "not gpu" -p no:cacheprovider 2>&1 | tail -4; \
	  echo "sanitizer reports: $$(ls build/asan/report.* 2>/dev/null | wc -l)"; cat build/asan/report.* 2>/dev/null | head -40 ) | tee profiles/r06_asan_tests.txt

build/fuzz/fuzz_host: tests/fuzz/fuzz_host.cpp build/asan/libhypergreco.so
	mkdir -p build/fuzz
	$(CLANG) -O1 -g -std=c++17 -fsanitize=fuzzer,address,undefined -mllvm -asan-globals=0 -shared-libsan tests/fuzz/fuzz_host.cpp -o $@ -Lbuild/asan -lhypergreco -Wl,-rpath,$(ROOT)/build/asan -Wl,-rpath,$(dir $(ASAN_RT))

fuzz: build/fuzz/fuzz_host
	$(MAKE) -C oracle
	python tests/fuzz/make_seeds.py
	mkdir -p profiles
	for t in json verify verifybn; do \
	  HG_FUZZ_TARGET=$$t HG_FUZZ_ROOT=$(ROOT) ASAN_OPTIONS=detect_leaks=0:verify_asan_link_order=0:detect_odr_violation=0 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 ./build/fuzz/fuzz_host build/fuzz/$$t -max_len=400000 -rss_limit_mb=6000 -timeout=60 \
	    -max_total_time=$(FUZZ_SECONDS) -print_final_stats=1 -artifact_prefix=build/fuzz/crash_$$t- > build/fuzz/$$t.log 2>&1; \
	  echo "exit code $$?" >> build/fuzz/$$t.log; \
	  ( echo "# make fuzz: HG_FUZZ_TARGET=$$t, $(FUZZ_SECONDS) s of libFuzzer (address + undefined-behaviour sanitizers) on tests/fuzz/fuzz_host.cpp"; grep -E "^#[0-9]+.*(INITED|DONE)|stat::|ERROR|SUMMARY|exit code" build/fuzz/$$t.log ) > profiles/r06_fuzz_$$t.txt; \
	done; cat profiles/r06_fuzz_*.txt

clean:
	$(MAKE) -C hyper-greco_amd/csrc clean
	rm -rf build/asan build/fuzz
.PHONY: all asan fuzz clean
