// Times svt_bam_evidence over dumped units (windows + breakpoints as the ctypes layer hands them over), reader alone (no HIP):
//   g++ -std=c++17 -O2 -I svtyper_amd/csrc -I include -o reader_time tools/reader_time.cpp svtyper_amd/csrc/svt_reads.cpp -lz -ldl -lpthread
//   reader_time <bam> <windows.bin> <breakpoints.bin> <threads> <repetitions> <read group>=<library index> ...
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <chrono>
#include "svtyper_reads.h"
#include "svt_error.h"
extern "C" const char* svt_last_error(void) { return svt::g_err.c_str(); }
template <class T> static std::vector<T> slurp(const char* path){ std::vector<T> v; if (FILE* f = std::fopen(path, "rb")) { std::fseek(f,0,SEEK_END); long n=std::ftell(f); std::fseek(f,0,SEEK_SET); v.resize((size_t)n/sizeof(T)); if(std::fread(v.data(),sizeof(T),v.size(),f)!=v.size()) v.clear(); std::fclose(f);} return v; }
int main(int argc, char** argv){
    svt_bam* bam=nullptr; if (svt_bam_open(argv[1], &bam)) return 3;
    auto win=slurp<svt_fetch_unit>(argv[2]); auto bps=slurp<svt_breakpoint>(argv[3]);
    int threads=atoi(argv[4]); int reps=atoi(argv[5]);
    std::vector<std::string> names; std::vector<int32_t> libs;
    for (int i=6;i<argc;++i){ const char* eq=strrchr(argv[i],'='); names.emplace_back(argv[i],(size_t)(eq-argv[i])); libs.push_back(atoi(eq+1)); }
    std::vector<const char*> np; for (auto& s:names) np.push_back(s.c_str());
    svt_summarise_args a{}; a.n_units=win.size(); a.windows=win.data(); a.breakpoints=bps.data(); a.n_read_groups=names.size(); a.read_groups=np.data(); a.read_group_lib=libs.data(); a.max_reads=1000; a.count_mode=1; a.n_threads=threads;
    std::vector<double> flank(1, 400.0); svt_evidence_params g{}; g.n_libs=1; g.lib_flank=flank.data(); g.min_aligned=20; g.split_slop=3;
    for (int r=0;r<reps;++r){ auto t0=std::chrono::steady_clock::now(); svt_evidence e{}; int rc=svt_bam_evidence(bam,&a,&g,&e); if(rc){printf("rc %d %s\n",rc,svt_last_error());return 7;}
      double ms=std::chrono::duration<double,std::milli>(std::chrono::steady_clock::now()-t0).count(); printf("%d threads: %zu units %llu records %.1f ms\n",threads,win.size(),(unsigned long long)e.rec_offset[win.size()],ms); svt_evidence_free(&e);}
    return 0; }
