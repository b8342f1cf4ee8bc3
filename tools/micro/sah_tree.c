// offline estimate: SAH cost of a binary BVH built by (a) Morton-order median-bit splits (LBVH topology) and (b) binned SAH top-down
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <stdint.h>
typedef struct { float lo[3], hi[3]; } Box;
static int N; static Box* tb; static float (*cen)[3];
static inline void grow(Box* a, const Box* b){ for(int k=0;k<3;k++){ if(b->lo[k]<a->lo[k])a->lo[k]=b->lo[k]; if(b->hi[k]>a->hi[k])a->hi[k]=b->hi[k]; } }
static inline Box empty(){ Box b; for(int k=0;k<3;k++){b.lo[k]=1e30f;b.hi[k]=-1e30f;} return b; }
static inline double area(const Box* b){ double d[3]; for(int k=0;k<3;k++) d[k]=b->hi[k]-b->lo[k]; if(d[0]<0) return 0; return d[0]*d[1]+d[1]*d[2]+d[2]*d[0]; }
static double cost_nodes, cost_leaves; static long nleaf, nnode; static int LEAF=2;
// ---- binned SAH
static int* LR; static int nint;
static int build_sah_id(int* idx,int n);
static Box build_sah(int* idx, int n){
  Box b=empty(), cb=empty(); for(int i=0;i<n;i++){ grow(&b,&tb[idx[i]]); Box c; for(int k=0;k<3;k++) c.lo[k]=c.hi[k]=cen[idx[i]][k]; grow(&cb,&c);} 
  if(n<=LEAF){ cost_leaves+=area(&b)*n; nleaf++; return b; }
  nnode++; cost_nodes+=area(&b);
  enum{NB=32}; double best=1e300; int bax=-1,bsp=0;
  for(int ax=0;ax<3;ax++){ float lo=cb.lo[ax], ext=cb.hi[ax]-lo; if(!(ext>0)) continue; Box bins[NB]; int cnt[NB]; for(int j=0;j<NB;j++){bins[j]=empty();cnt[j]=0;}
    for(int i=0;i<n;i++){ int j=(int)((cen[idx[i]][ax]-lo)/ext*NB); if(j>=NB)j=NB-1; grow(&bins[j],&tb[idx[i]]); cnt[j]++; }
    double ra[NB]; int rc[NB]; Box r=empty(); int c=0; for(int j=NB-1;j>0;j--){ grow(&r,&bins[j]); c+=cnt[j]; ra[j]=area(&r); rc[j]=c; }
    Box l=empty(); c=0; for(int j=0;j<NB-1;j++){ grow(&l,&bins[j]); c+=cnt[j]; if(c==0||rc[j+1]==0) continue; double cs=area(&l)*c+ra[j+1]*rc[j+1]; if(cs<best){best=cs;bax=ax;bsp=j;} }
  }
  int m;
  if(bax<0){ m=n/2; }
  else { float lo=cb.lo[bax], ext=cb.hi[bax]-lo; int i=0,j=n-1; while(i<=j){ int b_=(int)((cen[idx[i]][bax]-lo)/ext*NB); if(b_>=NB)b_=NB-1; if(b_<=bsp) i++; else { int t=idx[i]; idx[i]=idx[j]; idx[j]=t; j--; } } m=i; if(m==0||m==n) m=n/2; }
  build_sah(idx,m); build_sah(idx+m,n-m); return b;
}
// ---- LBVH (sorted morton, split at highest differing bit)
static uint64_t* code; 
static int cmpc(const void* a,const void* b){ uint64_t x=code[*(const int*)a], y=code[*(const int*)b]; return x<y?-1:x>y; }
static uint64_t expand(uint32_t v){ uint64_t x=v&0x1fffff; x=(x|x<<32)&0x1f00000000ffffULL; x=(x|x<<16)&0x1f0000ff0000ffULL; x=(x|x<<8)&0x100f00f00f00f00fULL; x=(x|x<<4)&0x10c30c30c30c30c3ULL; x=(x|x<<2)&0x1249249249249249ULL; return x; }
static Box build_lbvh(int* idx,int n,int bit){
  Box b=empty(); for(int i=0;i<n;i++) grow(&b,&tb[idx[i]]);
  if(n<=LEAF){ cost_leaves+=area(&b)*n; nleaf++; return b; }
  nnode++; cost_nodes+=area(&b);
  int m=-1; while(bit>=0){ uint64_t mask=1ULL<<bit; if(((code[idx[0]]^code[idx[n-1]])&mask)){ int lo=0,hi=n-1; while(lo<hi){int mid=(lo+hi)/2; if(code[idx[mid]]&mask) hi=mid; else lo=mid+1;} m=lo; break;} bit--; }
  if(m<=0||m>=n){ m=n/2; }
  build_lbvh(idx,m,bit-1); build_lbvh(idx+m,n-m,bit-1); return b;
}
// returns child code: >=0 internal id, <0 ~prim
static int build_sah_id(int* idx,int n){
  if(n==1) return ~idx[0];
  int id=nint++;
  Box cb=empty(); for(int i=0;i<n;i++){ Box c; for(int k=0;k<3;k++) c.lo[k]=c.hi[k]=cen[idx[i]][k]; grow(&cb,&c);}
  enum{NB=32}; double best=1e300; int bax=-1,bsp=0;
  for(int ax=0;ax<3;ax++){ float lo=cb.lo[ax], ext=cb.hi[ax]-lo; if(!(ext>0)) continue; Box bins[NB]; int cnt[NB]; for(int j=0;j<NB;j++){bins[j]=empty();cnt[j]=0;}
    for(int i=0;i<n;i++){ int j=(int)((cen[idx[i]][ax]-lo)/ext*NB); if(j>=NB)j=NB-1; grow(&bins[j],&tb[idx[i]]); cnt[j]++; }
    double ra[NB]; int rc[NB]; Box r=empty(); int c=0; for(int j=NB-1;j>0;j--){ grow(&r,&bins[j]); c+=cnt[j]; ra[j]=area(&r); rc[j]=c; }
    Box l=empty(); c=0; for(int j=0;j<NB-1;j++){ grow(&l,&bins[j]); c+=cnt[j]; if(c==0||rc[j+1]==0) continue; double cs=area(&l)*c+ra[j+1]*rc[j+1]; if(cs<best){best=cs;bax=ax;bsp=j;} }
  }
  int m;
  if(bax<0){ m=n/2; }
  else { float lo=cb.lo[bax], ext=cb.hi[bax]-lo; int i=0,j=n-1; while(i<=j){ int b_=(int)((cen[idx[i]][bax]-lo)/ext*NB); if(b_>=NB)b_=NB-1; if(b_<=bsp) i++; else { int t=idx[i]; idx[i]=idx[j]; idx[j]=t; j--; } } m=i; if(m==0||m==n) m=n/2; }
  int l=build_sah_id(idx,m); int r=build_sah_id(idx+m,n-m); LR[2*id]=l; LR[2*id+1]=r; return id;
}
int main(int argc,char**argv){ FILE*f=fopen(argv[1],"rb"); fseek(f,0,SEEK_END); long sz=ftell(f); fseek(f,0,SEEK_SET); N=sz/36; float* v=malloc(sz); fread(v,1,sz,f); fclose(f); if(argc>2) LEAF=atoi(argv[2]);
  tb=malloc(sizeof(Box)*N); cen=malloc(sizeof(float)*3*N); Box all=empty();
  for(int i=0;i<N;i++){ Box b=empty(); for(int j=0;j<3;j++){ Box p; for(int k=0;k<3;k++) p.lo[k]=p.hi[k]=v[i*9+j*3+k]; grow(&b,&p);} tb[i]=b; for(int k=0;k<3;k++) cen[i][k]=0.5f*(b.lo[k]+b.hi[k]); grow(&all,&b);} 
  int* idx=malloc(sizeof(int)*N); double A=area(&all);
  for(int i=0;i<N;i++) idx[i]=i; cost_nodes=cost_leaves=0; nleaf=nnode=0; build_sah(idx,N);
  printf("binned SAH : nodes %.3f leaves %.3f  (n %ld, leaves %ld)\n",cost_nodes/A,cost_leaves/A,nnode,nleaf);
  code=malloc(8*N); for(int i=0;i<N;i++){ uint32_t q[3]; for(int k=0;k<3;k++){ double t=(cen[i][k]-all.lo[k])/(all.hi[k]-all.lo[k]+1e-30); q[k]=(uint32_t)(t*2097151.0);} code[i]=expand(q[0])<<2|expand(q[1])<<1|expand(q[2]); }
  for(int i=0;i<N;i++) idx[i]=i; qsort(idx,N,sizeof(int),cmpc); cost_nodes=cost_leaves=0; nleaf=nnode=0; build_lbvh(idx,N,62);
  printf("LBVH       : nodes %.3f leaves %.3f  (n %ld, leaves %ld)\n",cost_nodes/A,cost_leaves/A,nnode,nleaf);
  if(argc>3){ for(int i=0;i<N;i++) idx[i]=i; LR=malloc(8*(size_t)N); nint=0; build_sah_id(idx,N); FILE*o=fopen(argv[3],"wb"); fwrite(&N,4,1,o); fwrite(LR,4,2*(size_t)(N-1),o); fclose(o); printf("wrote %s (%d internal)\n",argv[3],nint);} 
  return 0; }
