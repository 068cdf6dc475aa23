# Builds the HIP C-ABI library (gfx950) and the oracle's C helpers.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH  ?= gfx950
CSRC  := blackbox_amd/csrc
SRCS  := $(CSRC)/bbx_ctx.hip $(CSRC)/bbx_overscan.hip $(CSRC)/bbx_calibrate.hip \
         $(CSRC)/bbx_mask.hip $(CSRC)/bbx_select.hip $(CSRC)/bbx_lacosmic.hip $(CSRC)/bbx_xtalk.hip \
         $(CSRC)/bbx_stack.hip $(CSRC)/bbx_bkg.hip $(CSRC)/bbx_zogy.hip $(CSRC)/bbx_zogy3.hip $(CSRC)/bbx_sat.hip $(CSRC)/bbx_canny.hip $(CSRC)/bbx_psf.hip $(CSRC)/bbx_fpack.hip $(CSRC)/bbx_coadd.hip $(CSRC)/bbx_clipstats.hip
OBJS  := $(SRCS:.hip=.o)
LIB   := blackbox_amd/libbbx_hip.so
HOSTLIB := blackbox_amd/libbbx_host.so
# -ffp-contract=off: results must match numpy's unfused float32/float64 arithmetic
HIPFLAGS := --offload-arch=$(ARCH) -O3 -fPIC -ffp-contract=off -std=c++17 -Wall -Wno-unused-function

ORACLELIB := oracle/liblacosmic_c.so

all: $(LIB) $(HOSTLIB) $(ORACLELIB)

# the oracle's C twin of LA-Cosmic (test infrastructure: tests/ and bench.py's cpu_baseline leg only)
$(ORACLELIB): oracle/lacosmic_c.c
	gcc -O3 -fopenmp -ffp-contract=off -fPIC -shared -o $@ $< -lm

# host-side C helpers of the overscan solve (same float operations as the numpy code)
$(HOSTLIB): blackbox_amd/chost/bbx_host.c
	gcc -O3 -fPIC -shared -ffp-contract=off -fno-trapping-math -fno-math-errno -o $@ $< -lm

$(CSRC)/bbx_fft_gen.h: tools/gen_fft.py
	python3 tools/gen_fft.py

$(CSRC)/%.o: $(CSRC)/%.hip $(CSRC)/bbx_common.h $(CSRC)/bbx_mednet.h $(CSRC)/bbx_bsel.h $(CSRC)/bbx_fft_gen.h include/bbx.h
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(LIB): $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS) -L/opt/rocm/lib -lrocfft -Wl,-rpath,/opt/rocm/lib

clean:
	rm -f $(OBJS) $(LIB) $(HOSTLIB) $(ORACLELIB)

.PHONY: all clean
