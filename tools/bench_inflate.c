#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <zlib.h>
#include <bscall_amd.h>
static double now(void){struct timespec t;clock_gettime(CLOCK_MONOTONIC,&t);return t.tv_sec+1e-9*t.tv_nsec;}
int main(int argc,char**argv){
  FILE*f=fopen(argv[1],"rb");fseek(f,0,SEEK_END);long n=ftell(f);fseek(f,0,SEEK_SET);unsigned char*b=malloc(n);fread(b,1,n,f);fclose(f);
  static unsigned char out[65536+64];
  for(int mode=0;mode<4;mode++){
    double t0=now();unsigned long long tot=0;long p=0;int nb=0;
    while(p<n){unsigned xlen=b[p+10]|b[p+11]<<8;unsigned bs=(b[p+16]|b[p+17]<<8)+1;const unsigned char*pl=b+p+12+xlen;unsigned clen=bs-12-xlen-8;unsigned isize=*(unsigned*)(b+p+bs-4);unsigned crc=*(unsigned*)(b+p+bs-8);
      if(mode==0){z_stream z;memset(&z,0,sizeof z);inflateInit2(&z,-15);z.next_in=(Bytef*)pl;z.avail_in=clen;z.next_out=out;z.avail_out=65536;inflate(&z,Z_FINISH);inflateEnd(&z);}
      else if(mode==1){if(bsc_inflate_raw(pl,clen,out,isize)){printf("FAIL\n");return 1;}}
      else if(mode==2){crc32(crc32(0,0,0),out,isize);}
      else {if(bsc_crc32(out,isize)==0xdeadbeef)printf("x");}
      (void)crc;tot+=isize;p+=bs;nb++;}
    double dt=now()-t0;printf("%s: %d blocks %.3f s %.1f MB/s\n",mode==0?"zlib inflate":mode==1?"fast inflate":mode==2?"zlib crc32":"fast crc32",nb,dt,tot/dt/1e6);
  }
  return 0;}
