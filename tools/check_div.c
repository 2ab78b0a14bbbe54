/* Checks the Markstein constant-divisor step used by div_ln10_dev (kernels.hip) against IEEE division.
 * build: gcc -O2 -mfma -ffp-contract=off -o /tmp/check_div tools/check_div.c -lm ; run: /tmp/check_div 500000000 <seed> */
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <math.h>
#include <stdlib.h>
#define LN10 2.30258509299404568402
static inline uint64_t rotl(uint64_t x,int k){return (x<<k)|(x>>(64-k));}
static uint64_t s[4];
static uint64_t nxt(void){uint64_t r=rotl(s[1]*5,7)*9,t=s[1]<<17;s[2]^=s[0];s[3]^=s[1];s[1]^=s[2];s[0]^=s[3];s[2]^=t;s[3]=rotl(s[3],45);return r;}
int main(int argc,char**argv){
  long n=atol(argv[1]); s[0]=1;s[1]=2;s[2]=3;s[3]=atol(argv[2]);
  volatile double c=LN10; const double rc=1.0/c; /* correctly rounded reciprocal */
  long bad=0;
  for(long i=0;i<n;i++){
    uint64_t u=nxt(); double x;
    int mode=i&3;
    if(mode==0){ /* random mantissa, exponent in [-60,12] */
      uint64_t m=u&0xfffffffffffffULL; int e=(int)((u>>52)%73)-60; uint64_t b=((uint64_t)(e+1023)<<52)|m|(u&0x8000000000000000ULL); memcpy(&x,&b,8);
    } else if(mode==1){ x=-(double)(u>>11)*0x1p-53*2000.0; }
    else if(mode==2){ /* multiples of c near representable quotients: x = RN(q*c) +- few ulps */
      double q=(double)(u>>40)*0x1p-12; x=q*c; uint64_t b; memcpy(&b,&x,8); b+=(int)((u>>8)&7)-3; memcpy(&x,&b,8);
    } else { uint64_t b=u&0x7fefffffffffffffULL; if((b>>52)<2||(b>>52)>2040) b=0x3ff0000000000000ULL|(b&0xfffffffffffffULL); b|=u&0x8000000000000000ULL; memcpy(&x,&b,8);}
    double t=x/c;
    double q0=x*rc; double r=__builtin_fma(-c,q0,x); double q=__builtin_fma(r,rc,q0);
    if(memcmp(&t,&q,8)){ if(bad<10) printf("mismatch x=%a true=%a got=%a\n",x,t,q); bad++; }
  }
  printf("n=%ld bad=%ld rc=%a\n",n,bad,rc); return 0;}
