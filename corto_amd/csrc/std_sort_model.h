// std_sort_model.h — what libstdc++'s std::sort does to an array, as plain code that also compiles for the device.
//
// The reference orders a stream's (symbol, probability) list with std::sort and a comparator on the probability alone
// (src/tunstall.cpp:111-112), so the dictionary - and with it every byte of the block - depends on where std::sort leaves symbols
// of EQUAL probability.  std::sort is not stable and its permutation is a property of the algorithm, not of the standard: GCC's
// libstdc++ (the reference's and this repo's host toolchain) runs introsort - quicksort with the median of (first+1, middle,
// last-1) moved to the front as pivot and an unguarded Hoare partition, recursing on the upper part, until a range is 16
// elements or fewer or 2*floor(log2 n) levels have been used up (then heapsort) - and finishes with one insertion sort over
// the whole array.  This header restates those steps (bits/stl_algo.h: __introsort_loop, __move_median_to_first,
// __unguarded_partition, __final_insertion_sort; bits/stl_heap.h for the depth-limit fallback) so that the device builds the same
// list the host would.  tests/test_encode_stage_cpu.py compares it with std::sort itself on tie-heavy and adversarial inputs,
// with the depth limit lowered to force the heapsort branch as well.
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#define CRT_HD __host__ __device__
#else
#define CRT_HD
#endif

namespace corto_hip {

// T: trivially copyable; Less(a, b): strict weak order ("a goes before b")
template <typename T, typename Less>
struct StdSortModel {
	T *a; Less less;
	CRT_HD void swap(int i, int j) { const T t = a[i]; a[i] = a[j]; a[j] = t; }

	CRT_HD void median_to_first(int result, int x, int y, int z) {
		if(less(a[x], a[y])) {
			if(less(a[y], a[z])) swap(result, y);
			else if(less(a[x], a[z])) swap(result, z);
			else swap(result, x);
		} else if(less(a[x], a[z])) swap(result, x);
		else if(less(a[y], a[z])) swap(result, z);
		else swap(result, y);
	}
	CRT_HD int partition(int first, int last, int pivot) {
		for(;;) {
			while(less(a[first], a[pivot])) ++first;
			--last;
			while(less(a[pivot], a[last])) --last;
			if(!(first < last)) return first;
			swap(first, last);
			++first;
		}
	}
	// heap on a[first .. first+len): sift `value` down from hole, then up (std::__adjust_heap + __push_heap)
	CRT_HD void adjust_heap(int first, int hole, int len, T value) {
		const int top = hole;
		int child = hole;
		while(child < (len - 1)/2) {
			child = 2*(child + 1);
			if(less(a[first + child], a[first + child - 1])) child--;
			a[first + hole] = a[first + child];
			hole = child;
		}
		if((len & 1) == 0 && child == (len - 2)/2) {
			child = 2*(child + 1);
			a[first + hole] = a[first + child - 1];
			hole = child - 1;
		}
		int parent = (hole - 1)/2;
		while(hole > top && less(a[first + parent], value)) {
			a[first + hole] = a[first + parent];
			hole = parent;
			parent = (hole - 1)/2;
		}
		a[first + hole] = value;
	}
	CRT_HD void heap_sort(int first, int last) {            // std::__partial_sort(first, last, last): make_heap, then sort_heap
		const int len = last - first;
		if(len >= 2) for(int parent = (len - 2)/2;; parent--) { adjust_heap(first, parent, len, a[first + parent]); if(parent == 0) break; }
		for(int end = last; end - first > 1;) {
			--end;
			const T value = a[end];
			a[end] = a[first];
			adjust_heap(first, 0, end - first, value);
		}
	}
	CRT_HD void introsort(int first, int last, int depth) {
		// recursion on the upper part, iteration on the lower one; an explicit stack (ranges shrink, depth <= 2 log2 n)
		int stk_first[64], stk_last[64], stk_depth[64], sp = 0;
		for(;;) {
			while(last - first > 16) {
				if(depth == 0) { heap_sort(first, last); break; }
				--depth;
				const int mid = first + (last - first)/2;
				median_to_first(first, first + 1, mid, last - 1);
				const int cut = partition(first + 1, last, first);
				// upper part [cut, last) first (that is the order the recursion visits them in; the two parts are disjoint, so
				// the order does not change the result - kept anyway)
				stk_first[sp] = first; stk_last[sp] = cut; stk_depth[sp] = depth; sp++;
				first = cut;
			}
			if(sp == 0) break;
			--sp; first = stk_first[sp]; last = stk_last[sp]; depth = stk_depth[sp];
		}
	}
	CRT_HD void unguarded_linear_insert(int last) {
		const T val = a[last];
		int next = last - 1;
		while(less(val, a[next])) { a[last] = a[next]; last = next; --next; }
		a[last] = val;
	}
	CRT_HD void insertion_sort(int first, int last) {
		if(first == last) return;
		for(int i = first + 1; i != last; ++i) {
			if(less(a[i], a[first])) { const T val = a[i]; for(int k = i; k > first; k--) a[k] = a[k - 1]; a[first] = val; }
			else unguarded_linear_insert(i);
		}
	}
	CRT_HD void sort(int n, int depth_limit = -1) {
		if(n <= 0) return;
		int lg = 0; for(int m = n; m > 1; m >>= 1) lg++;
		introsort(0, n, depth_limit >= 0 ? depth_limit : 2*lg);
		if(n > 16) { insertion_sort(0, 16); for(int i = 16; i != n; ++i) unguarded_linear_insert(i); }
		else insertion_sort(0, n);
	}
};

template <typename T, typename Less>
CRT_HD inline void std_sort_model(T *a, int n, Less less, int depth_limit = -1) { StdSortModel<T, Less> m{a, less}; m.sort(n, depth_limit); }

} // namespace corto_hip
