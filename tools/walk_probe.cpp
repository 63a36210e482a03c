// experiment: order-free semantics ("global minimum, DFS order breaks exact ties") against the sequential walk
#include <cmath>
#include <cstdint>
#include <limits>
#include <thread>
#include <vector>
#include "../../root/repo/oracle/bvh.hpp"
using namespace orc;
namespace {
struct Walk {
  const MeshView &m1, &m2; M3 RT_R; V3 RT_T;
  double mind; int b1=-1,b2=-1; double relax; unsigned nbv=0, nleaf=0; unsigned near=0; double second=1e300;
  std::vector<std::pair<double,std::pair<int,int>>> leaves;
  Walk(const MeshView&a,const Tf&t1,const MeshView&b,const Tf&t2,double rl):m1(a),m2(b),relax(rl){ RT_R=tmul(t1.R,t2.R); RT_T=tmul(t1.R,t2.T-t1.T); mind=std::numeric_limits<double>::max(); }
  void leaf(int p1,int p2){ V3 S[3],T[3]; for(int k=0;k<3;++k){const double*p=m1.verts+3*size_t(m1.tris[3*p1+k]);const double*q=m2.verts+3*size_t(m2.tris[3*p2+k]);S[k]=V3(p[0],p[1],p[2]);T[k]=RT_R*V3(q[0],q[1],q[2])+RT_T;} V3 P,Q; double d=std::sqrt(sqr_tri_distance(S,T,P,Q)); ++nleaf; leaves.push_back({d,{p1,p2}}); if(mind>d){mind=d;b1=p1;b2=p2;} }
  bool stop(double c){ return c >= mind*(1+relax) && c>=mind+ (relax>0?1e-300:0); }
  void rec(unsigned i,unsigned j){ const hfcl_bvh_node&n1=m1.nodes[i],&n2=m2.nodes[j]; bool l1=n1.first_child<0,l2=n2.first_child<0; if(l1&&l2){leaf(-(n1.first_child+1),-(n2.first_child+1));return;}
    double s1=n1.obb_extent[0]*n1.obb_extent[0]+n1.obb_extent[1]*n1.obb_extent[1]+n1.obb_extent[2]*n1.obb_extent[2]; double s2=n2.obb_extent[0]*n2.obb_extent[0]+n2.obb_extent[1]*n2.obb_extent[1]+n2.obb_extent[2]*n2.obb_extent[2];
    unsigned a1,a2,c1,c2; if(l2||(!l1&&(s1>s2))){a1=n1.first_child;a2=j;c1=a1+1;c2=j;}else{a1=i;a2=n2.first_child;c1=i;c2=a2+1;}
    nbv+=2; double d1=rss_distance(RT_R,RT_T,m1.nodes[a1],m2.nodes[a2]),d2=rss_distance(RT_R,RT_T,m1.nodes[c1],m2.nodes[c2]);
    if(d2<d1){ if(!stop(d2))rec(c1,c2); if(!stop(d1))rec(a1,a2);} else { if(!stop(d1))rec(a1,a2); if(!stop(d2))rec(c1,c2);} }
};
Tf tfabi(const double* p){ return tf_from_abi(p); }
}
// out per query: [0]=seq d, [1]=relaxed d, [2]=seq b1,[3]=seq b2,[4]=rel b1,[5]=rel b2,[6]=#distinct leaves within 1e-12 rel of min (relaxed walk), [7]=nbv seq,[8]=nbv relaxed, [9]= #leaves with d==min exactly
extern "C" int tie_probe(const hfcl_bvh_node* nodes,const double* verts,const uint32_t* tris,const uint64_t* mt,size_t nm,const uint32_t* q1,const uint32_t* q2,const double* tf1,const double* tf2,size_t n,double* out,int nthreads){
  std::vector<MeshView> ms(nm); for(size_t i=0;i<nm;++i){ms[i].nodes=nodes+mt[4*i];ms[i].n_nodes=mt[4*i+1];ms[i].verts=verts+3*mt[4*i+2];ms[i].tris=tris+3*mt[4*i+3];}
  std::vector<std::thread> th; size_t chunk=(n+nthreads-1)/nthreads;
  for(int t=0;t<nthreads;++t) th.emplace_back([&,t]{ for(size_t i=t*chunk;i<std::min(n,(t+1)*chunk);++i){
    Tf a=tfabi(tf1+12*i),b=tfabi(tf2+12*i);
    Walk s(ms[q1[i]],a,ms[q2[i]],b,0.0); s.leaf(0,0); s.rec(0,0);
    Walk r(ms[q1[i]],a,ms[q2[i]],b,1e-9); r.leaf(0,0); r.rec(0,0);
    unsigned near=0,exact=0; for(auto&l:r.leaves){ if(l.first<=r.mind*(1+1e-12)) ++near; if(l.first==r.mind) ++exact; }
    double*o=out+10*i; o[0]=s.mind;o[1]=r.mind;o[2]=s.b1;o[3]=s.b2;o[4]=r.b1;o[5]=r.b2;o[6]=near;o[7]=s.nbv;o[8]=r.nbv;o[9]=exact; }});
  for(auto&x:th)x.join(); return 0; }
#include <cstdio>
namespace {
struct Trace { const MeshView &m1,&m2; M3 RT_R; V3 RT_T; double mind=1e300; int seq=0; int ta1,ta2,tb1,tb2; std::vector<double> chain;
  Trace(const MeshView&a,const Tf&t1,const MeshView&b,const Tf&t2):m1(a),m2(b){RT_R=tmul(t1.R,t2.R);RT_T=tmul(t1.R,t2.T-t1.T);}
  double leafd(int p1,int p2){V3 S[3],T[3];for(int k=0;k<3;++k){const double*p=m1.verts+3*size_t(m1.tris[3*p1+k]);const double*q=m2.verts+3*size_t(m2.tris[3*p2+k]);S[k]=V3(p[0],p[1],p[2]);T[k]=RT_R*V3(q[0],q[1],q[2])+RT_T;}V3 P,Q;return std::sqrt(sqr_tri_distance(S,T,P,Q));}
  void rec(unsigned i,unsigned j,double bound){ chain.push_back(bound); const hfcl_bvh_node&n1=m1.nodes[i],&n2=m2.nodes[j]; bool l1=n1.first_child<0,l2=n2.first_child<0; if(l1&&l2){int p1=-(n1.first_child+1),p2=-(n2.first_child+1); double d=leafd(p1,p2); ++seq; if((p1==ta1&&p2==ta2)||(p1==tb1&&p2==tb2)){ printf("leaf (%d,%d) seq %d d=%.17g mind_before=%.17g chain:",p1,p2,seq,d,mind); for(double c:chain)printf(" %.17g",c); printf("\n"); } if(d<mind)mind=d; chain.pop_back(); return;}
    double s1=n1.obb_extent[0]*n1.obb_extent[0]+n1.obb_extent[1]*n1.obb_extent[1]+n1.obb_extent[2]*n1.obb_extent[2]; double s2=n2.obb_extent[0]*n2.obb_extent[0]+n2.obb_extent[1]*n2.obb_extent[1]+n2.obb_extent[2]*n2.obb_extent[2];
    unsigned a1,a2,c1,c2; if(l2||(!l1&&(s1>s2))){a1=n1.first_child;a2=j;c1=a1+1;c2=j;}else{a1=i;a2=n2.first_child;c1=i;c2=a2+1;}
    double d1=rss_distance(RT_R,RT_T,m1.nodes[a1],m2.nodes[a2]),d2=rss_distance(RT_R,RT_T,m1.nodes[c1],m2.nodes[c2]);
    auto stop=[&](double c){return c>=mind*(1+1e-9);};
    if(d2<d1){ if(!stop(d2))rec(c1,c2,d2); if(!stop(d1))rec(a1,a2,d1);} else { if(!stop(d1))rec(a1,a2,d1); if(!stop(d2))rec(c1,c2,d2);} chain.pop_back(); }
};}
extern "C" int tie_trace(const hfcl_bvh_node* nodes,const double* verts,const uint32_t* tris,const uint64_t* mt,size_t nm,uint32_t q1,uint32_t q2,const double* tf1,const double* tf2,int ta1,int ta2,int tb1,int tb2){
  std::vector<MeshView> ms(nm); for(size_t i=0;i<nm;++i){ms[i].nodes=nodes+mt[4*i];ms[i].n_nodes=mt[4*i+1];ms[i].verts=verts+3*mt[4*i+2];ms[i].tris=tris+3*mt[4*i+3];}
  Trace t(ms[q1],tfabi(tf1),ms[q2],tfabi(tf2)); t.ta1=ta1;t.ta2=ta2;t.tb1=tb1;t.tb2=tb2; t.mind=t.leafd(0,0); t.rec(0,0,-1); printf("final mind %.17g\n",t.mind); return 0; }
// predictor experiment: state after `budget` steps of the explicit-stack walk vs total box tests
extern "C" int walk_predict(const hfcl_bvh_node* nodes,const double* verts,const uint32_t* tris,const uint64_t* mt,size_t nm,const uint32_t* q1,const uint32_t* q2,const double* tf1,const double* tf2,size_t n,int budget,double* out,int nthreads){
  std::vector<MeshView> ms(nm); for(size_t i=0;i<nm;++i){ms[i].nodes=nodes+mt[4*i];ms[i].n_nodes=mt[4*i+1];ms[i].verts=verts+3*mt[4*i+2];ms[i].tris=tris+3*mt[4*i+3];}
  std::vector<std::thread> th; size_t chunk=(n+nthreads-1)/nthreads;
  for(int t=0;t<nthreads;++t) th.emplace_back([&,t]{ for(size_t i=t*chunk;i<std::min(n,(t+1)*chunk);++i){
    const MeshView&m1=ms[q1[i]],&m2=ms[q2[i]]; Tf a=tfabi(tf1+12*i),b=tfabi(tf2+12*i);
    Walk w(m1,a,m2,b,0.0); w.leaf(0,0);
    struct Ent{unsigned i,j;double d;}; std::vector<Ent> st; st.push_back({0,0,-1}); int steps=0; double*o=out+6*i; bool rec=false; unsigned nbv=0;
    while(!st.empty()){
      if(!rec && steps>=budget){ rec=true; double mn=1e300; for(auto&e:st) if(e.d>=0) mn=std::min(mn,e.d); o[0]=w.mind; o[1]=mn; o[2]=st.size(); o[3]=nbv; }
      Ent e=st.back(); st.pop_back(); ++steps;
      if(e.d>=0 && e.d>=w.mind) continue;
      const hfcl_bvh_node&n1=m1.nodes[e.i],&n2=m2.nodes[e.j]; bool l1=n1.first_child<0,l2=n2.first_child<0;
      if(l1&&l2){ w.leaf(-(n1.first_child+1),-(n2.first_child+1)); continue; }
      double s1=n1.obb_extent[0]*n1.obb_extent[0]+n1.obb_extent[1]*n1.obb_extent[1]+n1.obb_extent[2]*n1.obb_extent[2]; double s2=n2.obb_extent[0]*n2.obb_extent[0]+n2.obb_extent[1]*n2.obb_extent[1]+n2.obb_extent[2]*n2.obb_extent[2];
      unsigned a1,a2,c1,c2; if(l2||(!l1&&(s1>s2))){a1=n1.first_child;a2=e.j;c1=a1+1;c2=e.j;}else{a1=e.i;a2=n2.first_child;c1=e.i;c2=a2+1;}
      nbv+=2; double d1=rss_distance(w.RT_R,w.RT_T,m1.nodes[a1],m2.nodes[a2]),d2=rss_distance(w.RT_R,w.RT_T,m1.nodes[c1],m2.nodes[c2]);
      if(d2<d1){ st.push_back({a1,a2,d1}); st.push_back({c1,c2,d2}); } else { st.push_back({c1,c2,d2}); st.push_back({a1,a2,d1}); }
    }
    if(!rec){o[0]=w.mind;o[1]=0;o[2]=0;o[3]=nbv;} o[4]=nbv; o[5]=w.mind; }});
  for(auto&x:th)x.join(); return 0; }
