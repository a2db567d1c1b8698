/* Thread-scaling check of the oracle's threaded port (used to find the elided page-touch bug):
 *   gcc -O2 -Ioracle oracle/mt_scaling_check.c oracle/hj_oracle.c -o /tmp/mt_check -lpthread -lm && /tmp/mt_check */
#include "hj_oracle.h"
#include <stdio.h>
#include <stdlib.h>
int main(){ uint64_t n=1<<26; uint64_t*R=malloc(n*8),*S=malloc(n*8); orc_generate_data("local_shuffle",n,n,1024,R); orc_generate_data("sorted",n,n,16,S);
 for(int t=1;t<=64;t*=4){ orc_result r; orc_build_probe_mt(R,n,S,n,4,64,t,0,&r); printf("threads %d build %.0f probe %.0f matches %llu\n",t,r.build_us,r.probe_us,(unsigned long long)r.totalMatches);} return 0; }
