#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <math.h>
#include <stdlib.h>
static inline uint64_t rotl(uint64_t x,int k){return (x<<k)|(x>>(64-k));}
static uint64_t s[4];
static uint64_t nxt(void){uint64_t r=rotl(s[1]*5,7)*9,t=s[1]<<17;s[2]^=s[0];s[3]^=s[1];s[1]^=s[2];s[0]^=s[3];s[2]^=t;s[3]=rotl(s[3],45);return r;}
int main(int argc,char**argv){
  long n=atol(argv[1]); s[0]=1;s[1]=2;s[2]=3;s[3]=atol(argv[2]);
  double kt[44]; for(int q=0;q<44;q++){double e=exp(-.1*q*2.30258509299404568402); if(e>.5)e=.5; kt[q]=e/(3.0-4.0*e);}
  long bad=0,tot=0;
  for(long i=0;i<n;i++){
    uint64_t u=nxt(),v=nxt();
    int sc=(int)(u&7); double mxd = sc<3?40: sc<6?400: (sc==6?6000:3e6);
    double x1=floor((double)((u>>8)&0xffffff)/16777216.0*mxd), x2=floor((double)((u>>32)&0xffffff)/16777216.0*mxd);
    if(x1+x2==0) continue;
    double k1=kt[(v&63)%44],k2=kt[((v>>6)&63)%44];
    double under=((v>>12)&3)==0?0.01:((v>>12)&3)==1?0.0:(double)((v>>16)&1023)/4096.0;
    double over=((v>>14)&3)==0?0.05:((v>>14)&3)==1?0.0:(double)((v>>26)&1023)/4096.0;
    double l=1.0-under,t=over; double lpt=l+t,lmt=l-t; if(!(lmt>0)) continue;
    double d=(x1+x2)*lmt;
    double num[3]={x1*(lpt+2.0*k2)-x2*(2.0-lpt+2.0*k1), x1*(2.0+lpt+4.0*k2)-x2*(2.0-lpt+4.0*k1), x1*(lpt+4.0*k2)-x2*(2.0-lpt+4.0*k1)};
    double y=1.0/d;
    for(int j=0;j<3;j++){ double tq=num[j]/d; double q0=num[j]*y; double r=__builtin_fma(-d,q0,num[j]); double q=__builtin_fma(r,y,q0); tot++;
      if(memcmp(&tq,&q,8)){ if(bad<10) printf("mismatch num=%a d=%a true=%a got=%a\n",num[j],d,tq,q); bad++; } }
  }
  printf("tot=%ld bad=%ld\n",tot,bad); return 0;}
