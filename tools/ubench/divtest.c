// does q' = fma(fma(-q,b,a), y, q) with y = RN(1/b), q = RN(a*y) equal RN(a/b)?  random + structured search
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <stdint.h>
#include <string.h>
static inline float fdiv_m(float a, float b, float y){ float q=a*y; float r=fmaf(-q,b,a); return fmaf(r,y,q);} 
int main(){ uint64_t s=88172645463325252ULL; long bad=0, n=0;
  for(long it=0; it<400000000L; ++it){ s^=s<<13; s^=s>>7; s^=s<<17; uint32_t ua=(uint32_t)s, ub=(uint32_t)(s>>32);
    // exponents restricted to avoid under/overflow: mantissa random, exponent in [-20,20]
    uint32_t ea=(ua>>23)%41+107, eb=(ub>>23)%41+107; ua=(ua&0x807FFFFF)|(ea<<23); ub=(ub&0x807FFFFF)|(eb<<23);
    float a,b; memcpy(&a,&ua,4); memcpy(&b,&ub,4); float y=1.0f/b; float q=fdiv_m(a,b,y); float t=a/b; n++;
    if(q!=t){ if(bad<10) printf("bad a=%a b=%a q=%a t=%a\n",a,b,q,t); bad++; } }
  printf("random: %ld bad of %ld\n",bad,n);
  // exhaustive over all mantissas of a for the Fenton constants
  float cs[]={0.065f,31.8364f,0.02f,3.33f,19.2f,160.0f,75.0f,6.24f,6.9f,17.0f,11.0f,6.8f,11.1f,8.5f,59.0f,17.54f,10.95f,7.44f,5.3f,9.6f,28.0f,16.0f,27.48f,5.0f,5.1237f,6.5f,9.0f,12.7f,13.0f,22.4f,60.0f,3.0f,8.0f,100.0f,180.0f,15.0f,0.00035f,1.367e-15f};
  for(unsigned k=0;k<sizeof cs/sizeof cs[0];++k){ float b=cs[k], y=1.0f/b; long bd=0; for(uint32_t m=0;m<(1u<<23);++m){ for(int e=100;e<=150;e+=10){ uint32_t u=m|((uint32_t)e<<23); float a; memcpy(&a,&u,4); if(fdiv_m(a,b,y)!=a/b) bd++; }} printf("c=%g bad=%ld\n",b,bd);} 
  return 0; }
